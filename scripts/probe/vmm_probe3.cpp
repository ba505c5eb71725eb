// Development probe 3: stability of (a) growing inside one big reservation, (b) remapping into a larger one.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static int fails = 0;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { ++fails; fprintf(stderr, "FAIL %-60s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
__global__ void touch(unsigned* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main(int argc, char** argv) {
    const bool remap = argc > 1 && !strcmp(argv[1], "remap");
    const bool sync_before = argc > 2 && !strcmp(argv[2], "sync");
    const bool whole = argc > 3 && !strcmp(argv[3], "whole");
    const bool sync_each = argc > 4 && !strcmp(argv[4], "syncmap");
    size_t g = 2 << 20;
    auto R = [&](size_t v) { return (v + g - 1) / g * g; };
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned; p.location.type = hipMemLocationTypeDevice; p.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = p.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    struct C { hipMemGenericAllocationHandle_t h; size_t off, size; };
    for (int rep = 0; rep < 20; ++rep) {
        std::vector<C> ch; char* base = nullptr; size_t va_bytes = 0, bytes = 0;
        va_bytes = remap ? R(64 << 20) : R((size_t)8 << 30);
        CK(hipMemAddressReserve((void**)&base, va_bytes, 0, nullptr, 0));
        for (int step = 1; step <= 12; ++step) {
            size_t need = R((size_t)step * 90 * 1000 * 1000);
            if (need > va_bytes) {
                if (sync_before) CK(hipDeviceSynchronize());
                char* va = nullptr; size_t vb = R(2 * need);
                CK(hipMemAddressReserve((void**)&va, vb, 0, nullptr, 0));
                for (auto& c : ch) { CK(hipMemUnmap(base + c.off, c.size)); CK(hipMemMap(va + c.off, c.size, 0, c.h, 0)); CK(hipMemSetAccess(va + c.off, c.size, &acc, 1)); }
                CK(hipMemAddressFree(base, va_bytes)); base = va; va_bytes = vb;
            }
            C c{}; c.off = bytes; c.size = need - bytes;
            if (sync_each) CK(hipDeviceSynchronize());
            CK(hipMemCreate(&c.h, c.size, &p, 0)); CK(hipMemMap(base + c.off, c.size, 0, c.h, 0));
            if (whole) CK(hipMemSetAccess(base, bytes + c.size, &acc, 1)); else CK(hipMemSetAccess(base + c.off, c.size, &acc, 1));
            ch.push_back(c); bytes += c.size;
            CK(hipMemsetAsync(base + c.off, 0, c.size, 0));
            size_t n = bytes / 4;
            touch<<<(unsigned)((n + 255) / 256), 256>>>((unsigned*)base, n);
            CK(hipGetLastError());
        }
        CK(hipDeviceSynchronize());
        unsigned first = 0, last = 0;
        CK(hipMemcpy(&first, base, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&last, base + bytes - 4, 4, hipMemcpyDeviceToHost));
        if (first != 12 || last != 1) { ++fails; fprintf(stderr, "rep %d: content first %u (12) last %u (1)\n", rep, first, last); }
        for (auto& c : ch) { CK(hipMemUnmap(base + c.off, c.size)); CK(hipMemRelease(c.h)); }
        CK(hipMemAddressFree(base, va_bytes));
    }
    fprintf(stderr, "%s%s: 20 reps x 12 grows, fails %d\n", remap ? "remap" : "in-place", sync_before ? "+sync" : "", fails);
    return fails != 0;
}
