// placement_vmm.hip — round-4 probe (NOT part of the product): does the WAY a 2 GiB batch buffer is
// allocated decide how fast a streaming kernel runs on it?  profiles/r03_micro.txt: the same launch
// measures 0.98 - 1.09 ms on different hipMalloc allocations of one process.  Here: N buffers by
// hipMalloc and N by the virtual-memory API (hipMemAddressReserve + hipMemCreate + hipMemMap, one
// physical handle per buffer / per 1 GiB / per 256 MiB chunk), each timed as the SOURCE and as the
// DESTINATION of a linear float4 copy against one fixed partner.
//
//   hipcc --offload-arch=gfx950 -O3 tools/placement_vmm.hip -o /tmp/placement_vmm && /tmp/placement_vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) copy4(const v4f* __restrict__ a, v4f* __restrict__ d, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) __builtin_nontemporal_store(a[i], d + i);
}
__global__ void fill(unsigned* p, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = (unsigned)i * 2654435761u;
}

static double time_copy(const void* a, void* d, size_t bytes) {
  const long n = bytes / 16;
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(copy4, dim3((n + 255) / 256), dim3(256), 0, 0, (const v4f*)a, (v4f*)d, n);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(copy4, dim3((n + 255) / 256), dim3(256), 0, 0, (const v4f*)a, (v4f*)d, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / reps;
}

static void* vmm_alloc(size_t bytes, size_t chunk, int dev) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (chunk % gran) chunk = (chunk + gran - 1) / gran * gran;
  size_t total = (bytes + chunk - 1) / chunk * chunk;
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, total, 1ull << 30, nullptr, 0));
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
    CK(hipMemRelease(h));
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, total, &acc, 1));
  return va;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 6;
  const size_t bytes = (size_t)64 * 3840 * 2160 * 4;
  int dev = 0;
  CK(hipSetDevice(dev));
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("allocation granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
  }
  void *pa, *pd;   // the fixed partners
  CK(hipMalloc(&pa, bytes)); CK(hipMalloc(&pd, bytes));
  hipLaunchKernelGGL(fill, dim3((bytes / 4 + 255) / 256), dim3(256), 0, 0, (unsigned*)pa, (long)(bytes / 4));
  CK(hipDeviceSynchronize());
  printf("partners: %p -> %p : %.1f us\n", pa, pd, time_copy(pa, pd, bytes));
  struct Kind { const char* name; size_t chunk; };
  const Kind kinds[] = {{"hipMalloc", 0}, {"vmm, one handle", bytes}, {"vmm, 1 GiB chunks", 1ull << 30},
                        {"vmm, 256 MiB chunks", 256ull << 20}, {"vmm, 2 MiB chunks", 2ull << 20}, {"hipMalloc again", 0}};
  for (const Kind& k : kinds) {
    std::vector<void*> bufs;
    printf("%-22s", k.name);
    for (int i = 0; i < N; i++) {
      void* p = nullptr;
      if (k.chunk == 0) CK(hipMalloc(&p, bytes));
      else p = vmm_alloc(bytes, k.chunk, dev);
      hipLaunchKernelGGL(fill, dim3((bytes / 4 + 255) / 256), dim3(256), 0, 0, (unsigned*)p, (long)(bytes / 4));
      CK(hipDeviceSynchronize());
      bufs.push_back(p);
      const double as_src = time_copy(p, pd, bytes), as_dst = time_copy(pa, p, bytes);
      printf("  [%p src %.0f dst %.0f]", p, as_src, as_dst);
      fflush(stdout);
    }
    printf("\n");
    // (kept allocated until the kind is done: a freed buffer would be handed out again)
    for (void* p : bufs) {
      if (k.chunk == 0) CK(hipFree(p));
      // (vmm buffers are left mapped: the process ends soon)
    }
  }
  return 0;
}
