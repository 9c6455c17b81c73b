// Developer micro-benchmark: duration of a (nearly) empty kernel as a function of the grid shape.
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor
// run under: rocprofv3 --kernel-trace --stats --output-format csv -d out -- /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TAG> __global__ void empty_kernel(const int *p, int *q) {
    if (p[blockIdx.x & 1023] == 123456789) q[threadIdx.x] = TAG;  // one scalar load per workgroup
}

template <int TAG> void run(int grid, int block, const int *p, int *q) {
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((empty_kernel<TAG>), dim3(grid), dim3(block), 0, 0, p, q);
    hipDeviceSynchronize();
}

int main() {
    int *p, *q;
    hipMalloc(&p, 4096);
    hipMalloc(&q, 4096);
    hipMemset(p, 0, 4096);
    run<0>(256, 256, p, q);
    run<1>(1024, 256, p, q);
    run<2>(1536, 256, p, q);
    run<3>(768, 512, p, q);
    run<4>(384, 1024, p, q);
    run<5>(6144, 64, p, q);
    run<6>(3072, 128, p, q);
    run<7>(3072, 256, p, q);
    printf("done\n");
    return 0;
}
