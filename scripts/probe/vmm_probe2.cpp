// Development probe 2: the engine's exact growth sequence, at a given rounding granularity (argv[1], bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
static int fails = 0;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { ++fails; printf("FAIL %-60s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    size_t g = argc > 1 ? strtoull(argv[1], 0, 10) : 4096;
    auto R = [&](size_t v) { return (v + g - 1) / g * g; };
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned; p.location.type = hipMemLocationTypeDevice; p.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = p.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    struct C { hipMemGenericAllocationHandle_t h; size_t off, size; };
    std::vector<C> ch; char* base = nullptr; size_t va_bytes = 0, bytes = 0;
    auto map_chunk = [&](size_t size) {
        C c{}; c.off = bytes; c.size = size;
        CK(hipMemCreate(&c.h, size, &p, 0)); CK(hipMemMap(base + c.off, size, 0, c.h, 0)); CK(hipMemSetAccess(base + c.off, size, &acc, 1));
        ch.push_back(c); bytes += size;
    };
    auto grow = [&](size_t want) {
        size_t need = R(want);
        if (!base) { va_bytes = R(std::max<size_t>(2 * need, 256ull << 20)); CK(hipMemAddressReserve((void**)&base, va_bytes, 0, nullptr, 0)); }
        if (need > va_bytes) {
            char* va = nullptr; size_t vb = R(2 * need);
            CK(hipMemAddressReserve((void**)&va, vb, 0, nullptr, 0));
            for (auto& c : ch) { CK(hipMemUnmap(base + c.off, c.size)); CK(hipMemMap(va + c.off, c.size, 0, c.h, 0)); CK(hipMemSetAccess(va + c.off, c.size, &acc, 1)); }
            CK(hipMemAddressFree(base, va_bytes)); base = va; va_bytes = vb;
            printf("  remapped %zu chunks into %zu bytes of VA\n", ch.size(), vb);
        }
        if (need > bytes) map_chunk(need - bytes);
        CK(hipMemset(base, 0x11, bytes)); CK(hipDeviceSynchronize());
        printf("grow to %zu: mapped %zu, chunks %zu, fails so far %d\n", want, bytes, ch.size(), fails);
    };
    for (size_t cap : {30000, 60000, 90000, 120000, 300000}) grow(cap * 3072);
    return fails != 0;
}
