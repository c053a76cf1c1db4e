"""Exception tree of the drop-in package.

Mirrors the names and message prefixes of the reference's ``kws/common/errors.py:6-63`` so callers'
``except`` clauses and log greps keep working: ``KWSError`` > ``DatasetError`` ("Dataset error: "),
``ModelError`` ("Model error: "), ``AudioProcessingError`` ("Audio processing error: ").
Codes coming back over the C ABI (``KWS_E*``) are turned into these by ``kws._native``.
"""
from __future__ import annotations

import logging
from typing import Any, Optional, Type

_log = logging.getLogger("kws")


class KWSError(Exception):
    """Root of every error raised by this package."""

    _prefix = ""
    _default = "An error occurred in KWS"

    def __init__(self, message: Optional[str] = None):
        text = self._default if message is None else message
        self.message = f"{self._prefix}{text}"
        super().__init__(self.message)


class DatasetError(KWSError):
    _prefix = "Dataset error: "
    _default = "Dataset error"


class ModelError(KWSError):
    _prefix = "Model error: "
    _default = "Model error"


class AudioProcessingError(KWSError):
    _prefix = "Audio processing error: "
    _default = "Audio processing error"


def handle_error(error: Exception, custom_error: Optional[Type[KWSError]] = None, msg: Optional[str] = None,
                 re_raise: bool = True) -> Any:
    """Log ``error`` and (by default) re-raise it, wrapped in ``custom_error`` when one is given.

    Same contract as the reference helper (``kws/common/errors.py:35-63``); must be called from
    inside an ``except`` block when ``custom_error`` is None so the bare ``raise`` has something to
    re-raise.  loguru is used when it is installed, the stdlib logger otherwise.
    """
    text = msg if msg else str(error)
    try:
        from loguru import logger as _loguru

        _loguru.error(f"{text}: {error}")
    except ImportError:
        _log.error("%s: %s", text, error)
    if not re_raise:
        return None
    if custom_error is not None:
        raise custom_error(text) from error
    raise error
