// Development probe: which hipMem* remap sequences does this ROCm accept?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); printf("%-70s -> %s\n", #x, hipGetErrorString(e)); } while (0)
int main() {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned; p.location.type = hipMemLocationTypeDevice; p.location.id = 0;
    size_t g = 0; CK(hipMemGetAllocationGranularity(&g, &p, hipMemAllocationGranularityRecommended));
    size_t gm = 0; CK(hipMemGetAllocationGranularity(&gm, &p, hipMemAllocationGranularityMinimum));
    printf("granularity recommended %zu minimum %zu\n", g, gm);
    hipMemAccessDesc acc = {}; acc.location = p.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t c1 = 48 * g, c2 = 16 * g;
    void* va = nullptr; CK(hipMemAddressReserve(&va, 64 * g, 0, nullptr, 0));
    hipMemGenericAllocationHandle_t h1, h2, h3;
    CK(hipMemCreate(&h1, c1, &p, 0)); CK(hipMemMap(va, c1, 0, h1, 0)); CK(hipMemSetAccess(va, c1, &acc, 1));
    CK(hipMemCreate(&h2, c2, &p, 0)); CK(hipMemMap((char*)va + c1, c2, 0, h2, 0)); CK(hipMemSetAccess((char*)va + c1, c2, &acc, 1));
    CK(hipMemset(va, 0x5A, c1 + c2)); CK(hipDeviceSynchronize());
    // remap into a larger range
    void* vb = nullptr; CK(hipMemAddressReserve(&vb, 256 * g, 0, nullptr, 0));
    printf("-- variant A: unmap old, map new, set access per chunk\n");
    CK(hipMemUnmap(va, c1)); CK(hipMemMap(vb, c1, 0, h1, 0)); CK(hipMemSetAccess(vb, c1, &acc, 1));
    CK(hipMemUnmap((char*)va + c1, c2)); CK(hipMemMap((char*)vb + c1, c2, 0, h2, 0)); CK(hipMemSetAccess((char*)vb + c1, c2, &acc, 1));
    printf("-- variant B: set access over the whole mapped range\n");
    CK(hipMemSetAccess(vb, c1 + c2, &acc, 1));
    CK(hipMemAddressFree(va, 64 * g));
    unsigned char probe[4] = {0, 0, 0, 0};
    CK(hipMemcpy(probe, (char*)vb + c1 + 5, 4, hipMemcpyDeviceToHost));
    printf("content after remap: %02x %02x (expect 5a)\n", probe[0], probe[3]);
    CK(hipMemCreate(&h3, c2, &p, 0)); CK(hipMemMap((char*)vb + c1 + c2, c2, 0, h3, 0)); CK(hipMemSetAccess((char*)vb + c1 + c2, c2, &acc, 1));
    CK(hipMemset((char*)vb + c1 + c2, 1, c2)); CK(hipDeviceSynchronize());
    printf("-- variant C: map the same handle at a second address while still mapped at the first\n");
    void* vc = nullptr; CK(hipMemAddressReserve(&vc, 256 * g, 0, nullptr, 0));
    CK(hipMemMap(vc, c1, 0, h1, 0)); CK(hipMemSetAccess(vc, c1, &acc, 1));
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot)); printf("free %zu total %zu\n", fr, tot);
    void* big = nullptr; CK(hipMemAddressReserve(&big, (size_t)288 << 30, 0, nullptr, 0));
    std::vector<void*> many; int okc = 0;
    for (int i = 0; i < 2000; ++i) { void* q = nullptr; if (hipMemAddressReserve(&q, (size_t)288 << 30, 0, nullptr, 0) != hipSuccess) break; many.push_back(q); ++okc; }
    printf("288 GiB reservations that succeeded: %d\n", okc);
    return 0;
}
