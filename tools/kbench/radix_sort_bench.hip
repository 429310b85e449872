#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <chrono>
#include <algorithm>
int main(int argc, char **argv) {
    size_t n = argc > 1 ? atol(argv[1]) : 2600000;
    int end_bit = argc > 2 ? atoi(argv[2]) : 52;
    std::vector<unsigned long long> h(n);
    unsigned long long x = 88172645463325252ull;
    for (auto &k : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; k = x & ((1ull << end_bit) - 1); }
    unsigned long long *ki, *ko; float *vi, *vo;
    hipMalloc(&ki, n * 8); hipMalloc(&ko, n * 8); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
    hipMemcpy(ki, h.data(), n * 8, hipMemcpyHostToDevice);
    size_t tb = 0;
    rocprim::radix_sort_pairs(nullptr, tb, ki, ko, vi, vo, n, 0, end_bit, 0);
    void *tmp; hipMalloc(&tmp, tb);
    for (int r = 0; r < 5; ++r) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        rocprim::radix_sort_pairs(tmp, tb, ki, ko, vi, vo, n, 0, end_bit, 0);
        hipDeviceSynchronize();
        printf("n %zu bits %d temp %zu: %.1f us\n", n, end_bit, tb, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::vector<unsigned long long> o(n);
    hipMemcpy(o.data(), ko, n * 8, hipMemcpyDeviceToHost);
    printf("sorted %d\n", (int)std::is_sorted(o.begin(), o.end()));
}
