// Micro-benchmark (GPU box): host-observed latency of launch -> completion for a trivial kernel, launched eagerly and as a
// replayed hipGraph of 1, 2 and 4 kernel nodes, with hipStreamSynchronize and with a hipStreamQuery spin as the wait.
// Answers VERDICT r02 item 4's question: why is the streaming push slower as a hipGraph (one kernel node) than eager?
//   hipcc --offload-arch=gfx950 -O2 tools/graph_overhead.hip -o tools/bin/graph_overhead && tools/bin/graph_overhead
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void tiny(int* p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}

static double p50(std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    int* d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 3000;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto wait = [&](bool spin) {
        if (spin) {
            while (hipStreamQuery(s) == hipErrorNotReady) {
            }
        } else {
            hipStreamSynchronize(s);
        }
    };
    for (int spin = 0; spin < 2; ++spin) {
        for (int nodes : {1, 2, 4}) {
            std::vector<double> eager, graph;
            for (int i = 0; i < N; ++i) {
                const double t0 = now();
                for (int k = 0; k < nodes; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d);
                wait(spin);
                eager.push_back(now() - t0);
            }
            hipGraph_t g;
            hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            for (int k = 0; k < nodes; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d);
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int i = 0; i < N; ++i) {
                const double t0 = now();
                hipGraphLaunch(ge, s);
                wait(spin);
                graph.push_back(now() - t0);
            }
            // host time of the launch call alone (no wait)
            std::vector<double> el, gl;
            for (int i = 0; i < 500; ++i) {
                double t0 = now();
                for (int k = 0; k < nodes; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(64), 0, s, d);
                el.push_back(now() - t0);
                hipStreamSynchronize(s);
                t0 = now();
                hipGraphLaunch(ge, s);
                gl.push_back(now() - t0);
                hipStreamSynchronize(s);
            }
            printf("%s wait, %d kernel(s): eager p50 %.2f us (launch call %.2f), hipGraph replay p50 %.2f us (launch call %.2f)\n",
                   spin ? "hipStreamQuery spin" : "hipStreamSynchronize", nodes, p50(eager), p50(el), p50(graph), p50(gl));
            hipGraphExecDestroy(ge);
            hipGraphDestroy(g);
        }
    }
    return 0;
}
