// vmm_shim.hip — experiment helper (NOT part of the product): device buffers composed from
// physical chunks through HIP's virtual-memory API, so that tools/placement_vmm.py can ask what
// about an ALLOCATION makes the strip-shaped kernels 10 % faster or slower on it (round 5, review
// item 2).  Host code only.
//
//   hipcc -O2 -shared -fPIC tools/vmm_shim.hip -o tools/libvmm_shim.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "vmm_shim: HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

extern "C" {

// n buffers of `bytes` each, composed of chunks of `chunk` bytes.
// order 0: buffer after buffer, chunk after chunk (create + map in sequence)
// order 1: all chunks of all buffers created first (in sequence), then dealt to the buffers
//          ROUND-ROBIN (chunk i of the pool -> buffer i % n): the buffers interleave physically
// order 2: all chunks created first, then mapped in a RANDOM permutation
// order 3: all chunks created first, buffers take them in REVERSE order of creation
int vmm_alloc_many(int dev, size_t bytes, size_t chunk, int order, uint64_t seed, int n, void** out) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (chunk == 0 || chunk > bytes) chunk = bytes;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t per = (bytes + chunk - 1) / chunk, total = per * chunk;
  for (int b = 0; b < n; b++) CK(hipMemAddressReserve(&out[b], total, 1ull << 21, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(per * n);
  if (order == 0) {
    for (int b = 0; b < n; b++)
      for (size_t i = 0; i < per; i++) {
        CK(hipMemCreate(&h[b * per + i], chunk, &prop, 0));
        CK(hipMemMap((char*)out[b] + i * chunk, chunk, 0, h[b * per + i], 0));
      }
  } else {
    for (size_t i = 0; i < per * n; i++) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    std::vector<size_t> idx(per * n);
    for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
    if (order == 2) {
      rng_state = seed ? seed : 1;
      for (size_t i = idx.size() - 1; i > 0; i--) std::swap(idx[i], idx[rnd() % (i + 1)]);
    } else if (order == 3) {
      std::reverse(idx.begin(), idx.end());
    }
    for (size_t j = 0; j < idx.size(); j++) {
      int b; size_t i;
      if (order == 1) { b = (int)(j % n); i = j / n; }
      else { b = (int)(j / per); i = j % per; }
      CK(hipMemMap((char*)out[b] + i * chunk, chunk, 0, h[idx[j]], 0));
    }
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int b = 0; b < n; b++) CK(hipMemSetAccess(out[b], total, &acc, 1));
  for (auto& x : h) CK(hipMemRelease(x));   // the mappings keep the memory
  return 0;
}

int vmm_free(void* p, size_t bytes, size_t chunk) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  size_t gran = 4096;
  if (chunk == 0 || chunk > bytes) chunk = bytes;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t per = (bytes + chunk - 1) / chunk, total = per * chunk;
  CK(hipDeviceSynchronize());
  CK(hipMemUnmap(p, total));
  CK(hipMemAddressFree(p, total));
  return 0;
}

int vmm_mem_info(size_t* free_b, size_t* total_b) { CK(hipMemGetInfo(free_b, total_b)); return 0; }

}  // extern "C"
