// PMC calibration (MI355X_MICROARCH.md §HBM): how many bytes does FETCH_SIZE report for a streaming read of a known size
// at 4 B / lane (the access width of the FAST cell kernel's tile loads) and at 16 B / lane?  Buffers are 1 GiB (> 256 MiB L3).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const unsigned *p, size_t n, unsigned *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read16(const uint4 *p, size_t n, unsigned *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) { uint4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = 1ull << 30;
    unsigned *d, *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 64);
    hipMemset(d, 1, bytes);
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(read4, dim3(2048), dim3(256), 0, 0, d, bytes / 4, o);
        hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("read %zu bytes per launch\n", bytes);
    return 0;
}
