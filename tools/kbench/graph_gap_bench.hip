// What a HIP graph would buy the tail of a fused call: four small dependent kernels (the shape of scan -> re-score ->
// scatter -> rank at 1 Mbp) enqueued on a stream, replayed as an instantiated graph, and replayed after the arguments of
// every node were replaced (a call's thresholds, capacities and pointers change).  Wall time from the first enqueue to
// the end of the synchronisation, median of 400.
//   hipcc --offload-arch=gfx950 -O3 graph_gap_bench.hip -o graph_gap_bench && ./graph_gap_bench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                           \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

__global__ void step(unsigned *p, unsigned add)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        p[0] += add;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main()
{
    constexpr int kNodes = 4, kReps = 400;
    unsigned *d = nullptr;
    CHECK(hipMalloc(&d, 256));
    CHECK(hipMemset(d, 0, 256));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<double> a, b, c;
    for (int r = 0; r < kReps + 50; ++r) {  // (a) plain launches
        const double t0 = now_us();
        for (int i = 0; i < kNodes; ++i)
            hipLaunchKernelGGL(step, dim3(256), dim3(256), 0, st, d, 1u);
        CHECK(hipStreamSynchronize(st));
        if (r >= 50)
            a.push_back(now_us() - t0);
    }
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < kNodes; ++i)
        hipLaunchKernelGGL(step, dim3(256), dim3(256), 0, st, d, 1u);
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int r = 0; r < kReps + 50; ++r) {  // (b) the graph replayed unchanged
        const double t0 = now_us();
        CHECK(hipGraphLaunch(exec, st));
        CHECK(hipStreamSynchronize(st));
        if (r >= 50)
            b.push_back(now_us() - t0);
    }
    size_t nn = 0;
    CHECK(hipGraphGetNodes(graph, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CHECK(hipGraphGetNodes(graph, nodes.data(), &nn));
    for (int r = 0; r < kReps + 50; ++r) {  // (c) every node's arguments replaced before the replay
        unsigned add = (unsigned)r;
        unsigned *ptr = d;
        void *args[2] = {&ptr, &add};
        const double t0 = now_us();
        for (size_t i = 0; i < nn; ++i) {
            hipKernelNodeParams kp{};
            kp.func = reinterpret_cast<void *>(step);
            kp.gridDim = dim3(256);
            kp.blockDim = dim3(256);
            kp.sharedMemBytes = 0;
            kp.kernelParams = args;
            kp.extra = nullptr;
            CHECK(hipGraphExecKernelNodeSetParams(exec, nodes[i], &kp));
        }
        CHECK(hipGraphLaunch(exec, st));
        CHECK(hipStreamSynchronize(st));
        if (r >= 50)
            c.push_back(now_us() - t0);
    }
    std::printf("4 dependent small kernels, first enqueue -> synchronised, median of %d (us)\n", kReps);
    std::printf("  stream launches                          %7.1f\n", median(a));
    std::printf("  graph replay                             %7.1f\n", median(b));
    std::printf("  graph replay, node arguments replaced    %7.1f\n", median(c));
    return 0;
}
