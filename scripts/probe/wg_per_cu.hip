// How many workgroups of T threads does a CU hold at once?  Each workgroup spins for a fixed time; the launch's
// duration over that time is ceil(workgroups per CU requested / workgroups per CU resident).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS>
__global__ void spin(uint64_t ticks, uint32_t* out) {
    __shared__ uint32_t pad[LDS / 4];
    pad[threadIdx.x] = threadIdx.x;
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = pad[(ticks & 63)];
}
template <int LDS>
void run(int threads, int per_cu) {
    uint32_t* out;
    hipMalloc(&out, 256 * 64 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<LDS>, threads, 0);
    const uint64_t ticks = 100000;  // 1 ms at 100 MHz
    spin<LDS><<<256, threads>>>(ticks, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    spin<LDS><<<256 * per_cu, threads>>>(ticks, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("threads %d lds %d requested %d per CU: api says %d, launch took %.2f ms (1 ms per round)\n", threads, LDS, per_cu, occ, ms);
    hipFree(out);
}
int main() {
    for (int per_cu : {8, 10, 12, 16, 24, 32}) run<4096>(64, per_cu);
    for (int per_cu : {4, 6, 8, 12, 16}) run<8192>(128, per_cu);
    for (int per_cu : {8, 10, 11, 12}) run<14612>(64, per_cu);
    return 0;
}
