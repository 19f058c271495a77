// runtime.hip — context / memory / event half of the C ABI (include/imgproc_hip.h).
// One context per device: a dedicated non-blocking HIP stream, a grow-only
// staging workspace for the host-pointer entry points and a small table arena.
#include <stdarg.h>

#include "common.hpp"

static thread_local std::string g_tls_error;

void ipa_set_error(ipa_ctx* ctx, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->last_error = buf;
  g_tls_error = buf;
}

extern "C" {

int ipa_version(void) { return IPA_VERSION; }

const char* ipa_status_string(int s) {
  switch (s) {
    case IPA_OK: return "ok";
    case IPA_ERR_BAD_ARG: return "bad argument";
    case IPA_ERR_UNSUPPORTED: return "unsupported dtype/mode";
    case IPA_ERR_HIP: return "HIP runtime error";
    case IPA_ERR_OOM: return "out of device memory";
    case IPA_ERR_NO_DEVICE: return "no usable gfx950 device";
  }
  return "unknown status";
}

const char* ipa_last_error(const ipa_ctx* ctx) {
  return ctx ? ctx->last_error.c_str() : g_tls_error.c_str();
}

int ipa_device_count(int* count) {
  if (!count) return IPA_ERR_BAD_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    ipa_set_error(nullptr, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return IPA_ERR_NO_DEVICE;
  }
  *count = n;
  return IPA_OK;
}

namespace {
struct TuneName {
  const char* name;
  const char* env;
  int ipa_tuning::*field;
};
const TuneName kTuneNames[] = {
    {"strip_h", "IPA_STRIP_H", &ipa_tuning::strip_h},
    {"frames_inner", "IPA_FRAMES_INNER", &ipa_tuning::frames_inner},
    {"big_wave", "IPA_BIG_WAVE", &ipa_tuning::big_wave},
    {"big_fused", "IPA_BIG_FUSED", &ipa_tuning::big_fused},
    {"stream_k", "IPA_STREAM_K", &ipa_tuning::stream_k},
    {"ring_min", "IPA_RING_MIN", &ipa_tuning::ring_min},
    {"ring_remap", "IPA_RING_REMAP", &ipa_tuning::ring_remap},
    {"stored_coords", "IPA_STORED_COORDS", &ipa_tuning::stored_coords},
    {"tile_warp", "IPA_TILE_WARP", &ipa_tuning::tile_warp},
    {"lens_cache", "IPA_LENS_CACHE", &ipa_tuning::lens_cache},
    {"frames_wg", "IPA_FRAMES_WG", &ipa_tuning::frames_wg},
    {"frame_major", "IPA_FRAME_MAJOR", &ipa_tuning::frame_major},
    {"pipe7", "IPA_PIPE7", &ipa_tuning::pipe7},
    {"group_chunk", "IPA_GROUP_CHUNK", &ipa_tuning::group_chunk},
#if IPA_WITH_TILE_CHAIN
    {"tile_chain", "IPA_TILE_CHAIN", &ipa_tuning::tile_chain},
    {"chain_steps", "IPA_CHAIN_STEPS", &ipa_tuning::chain_steps},
    {"chain_frames", "IPA_CHAIN_FRAMES", &ipa_tuning::chain_frames},
#endif
    {"pipe", "IPA_PIPE_LOOPS", &ipa_tuning::pipe},
    {"u8_lz_lds", "IPA_U8_LZ_LDS", &ipa_tuning::u8_lz_lds},
    {"rank1_sep", "IPA_RANK1_SEP", &ipa_tuning::rank1_sep},
    {"tail_rows", "IPA_TAIL_ROWS", &ipa_tuning::tail_rows},
    {"sep_u16", "IPA_SEP_U16", &ipa_tuning::sep_u16},
    {"strip_remap", "IPA_STRIP_REMAP", &ipa_tuning::strip_remap},
};
}  // namespace

static bool tune_in_range(const char* name, int v) {
  if (strcmp(name, "strip_h") == 0) return v >= 0 && v <= 4096;
  if (strcmp(name, "stream_k") == 0) return v >= 7 && v <= 99;
  if (strcmp(name, "ring_min") == 0) return v >= 1;
  if (strcmp(name, "ring_remap") == 0) return v >= 0 && v <= 2;
  if (strcmp(name, "tile_warp") == 0) return v >= 0 && v <= 2;
  if (strcmp(name, "stored_coords") == 0) return v >= 0;
  if (strcmp(name, "group_chunk") == 0) return v >= -1 && v <= 4096;
  if (strcmp(name, "chain_steps") == 0) return v >= 0 && v <= 64;
  if (strcmp(name, "chain_frames") == 0) return v >= 0 && v <= 8;
  if (strcmp(name, "rank1_sep") == 0) return v >= 0 && v <= 3;
  if (strcmp(name, "tail_rows") == 0) return v >= -1 && v <= 4096;
  if (strcmp(name, "tile_chain") == 0) return v >= 0 && v <= 2;
  return v == 0 || v == 1;
}

int ipa_ctx_set_tuning(ipa_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return IPA_ERR_BAD_ARG;
  for (const TuneName& t : kTuneNames)
    if (strcmp(t.name, name) == 0) {
      if (!tune_in_range(name, value)) {
        ipa_set_error(ctx, "tuning knob '%s': value %d out of range", name, value);
        return IPA_ERR_BAD_ARG;
      }
      ctx->tune.*(t.field) = value;
      return IPA_OK;
    }
  ipa_set_error(ctx, "unknown tuning knob '%s'", name);
  return IPA_ERR_BAD_ARG;
}

int ipa_ctx_get_tuning(ipa_ctx* ctx, const char* name, int* value) {
  if (!ctx || !name || !value) return IPA_ERR_BAD_ARG;
#if IPA_WITH_TILE_CHAIN
  if (strcmp(name, "chain_launches") == 0) {   // read-only counter, not a knob
    *value = (int)(ctx->chain_launches & 0x7fffffff);
    return IPA_OK;
  }
#endif
  if (strcmp(name, "rank1_routed") == 0) {
    *value = (int)(ctx->rank1_routed & 0x7fffffff);
    return IPA_OK;
  }
  if (strcmp(name, "strip_remaps") == 0) {
    *value = (int)(ctx->strip_remaps & 0x7fffffff);
    return IPA_OK;
  }
  if (strcmp(name, "tail_rows_used") == 0) {
    *value = ctx->tail_rows_used;
    return IPA_OK;
  }
  if (strcmp(name, "group_chunk_used") == 0) {
    *value = ctx->group_chunk_used;
    return IPA_OK;
  }
  for (const TuneName& t : kTuneNames)
    if (strcmp(t.name, name) == 0) {
      *value = ctx->tune.*(t.field);
      return IPA_OK;
    }
  ipa_set_error(ctx, "unknown tuning knob '%s'", name);
  return IPA_ERR_BAD_ARG;
}

int ipa_ctx_create(int device_id, ipa_ctx** out) {
  if (!out) return IPA_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    ipa_set_error(nullptr, "no HIP device visible (this library has no CPU fallback)");
    return IPA_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= n) {
    ipa_set_error(nullptr, "device_id %d out of range [0,%d)", device_id, n);
    return IPA_ERR_BAD_ARG;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
    ipa_set_error(nullptr, "hipGetDeviceProperties failed");
    return IPA_ERR_HIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    ipa_set_error(nullptr, "device %d is %s; this library carries gfx950 code objects only",
                  device_id, prop.gcnArchName);
    return IPA_ERR_NO_DEVICE;
  }
  ipa_ctx* c = new ipa_ctx();
  // environment defaults of the tuning knobs: read here, once per context
  // (out-of-range values and knobs of kernels this build does not carry are ignored)
  for (const TuneName& t : kTuneNames)
    if (const char* e = getenv(t.env)) {
      const int v = atoi(e);
      if (tune_in_range(t.name, v))
        c->tune.*(t.field) = v;
    }
  c->device = device_id;
  c->cu_count = prop.multiProcessorCount;
  if (hipSetDevice(device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    ipa_set_error(nullptr, "stream creation failed on device %d", device_id);
    delete c;
    return IPA_ERR_HIP;
  }
  *out = c;
  return IPA_OK;
}

int ipa_ctx_destroy(ipa_ctx* c) {
  if (!c) return IPA_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->ws) (void)hipFree(c->ws);
  if (c->tab) (void)hipFree(c->tab);
  if (c->plan) (void)hipFree(c->plan);
  if (c->lens_map) (void)hipFree(c->lens_map);
  for (auto& hnt : c->tile_slow)
    if (hnt.copied) (void)hipEventDestroy(hnt.copied);
  if (c->tile_slow_dev) (void)hipFree(c->tile_slow_dev);
  if (c->tile_slow_host) (void)hipHostFree(c->tile_slow_host);
  if (c->ring_hint) (void)hipHostFree(c->ring_hint);
  if (c->tab_pinned) (void)hipHostFree(c->tab_pinned);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return IPA_OK;
}

int ipa_ctx_synchronize(ipa_ctx* c) {
  if (!c) return IPA_ERR_BAD_ARG;
  IPA_HIP(c, hipStreamSynchronize(c->stream));
  return IPA_OK;
}

int ipa_device_pci_bus_id(int device_id, char* buf, size_t len) {
  if (!buf || len < 16) return IPA_ERR_BAD_ARG;
  if (hipDeviceGetPCIBusId(buf, (int)len, device_id) != hipSuccess) return IPA_ERR_HIP;
  return IPA_OK;
}

int ipa_ctx_device_info(ipa_ctx* c, char* name, size_t name_len, int* cu_count,
                        size_t* total_mem) {
  if (!c) return IPA_ERR_BAD_ARG;
  hipDeviceProp_t prop;
  IPA_HIP(c, hipGetDeviceProperties(&prop, c->device));
  if (name && name_len) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (total_mem) *total_mem = prop.totalGlobalMem;
  return IPA_OK;
}

int ipa_mem_info(ipa_ctx* c, size_t* free_bytes, size_t* total_bytes) {
  if (!c || !free_bytes || !total_bytes) return IPA_ERR_BAD_ARG;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipMemGetInfo(free_bytes, total_bytes));
  return IPA_OK;
}

int ipa_malloc(ipa_ctx* c, size_t bytes, void** dptr) {
  if (!c || !dptr) return IPA_ERR_BAD_ARG;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipMalloc(dptr, bytes ? bytes : 1));
  return IPA_OK;
}

int ipa_free(ipa_ctx* c, void* dptr) {
  if (!c) return IPA_ERR_BAD_ARG;
  if (!dptr) return IPA_OK;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipStreamSynchronize(c->stream));
  IPA_HIP(c, hipFree(dptr));
  return IPA_OK;
}

int ipa_host_alloc(ipa_ctx* c, size_t bytes, void** hptr) {
  if (!c || !hptr) return IPA_ERR_BAD_ARG;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
  return IPA_OK;
}

int ipa_host_free(ipa_ctx* c, void* hptr) {
  if (!c) return IPA_ERR_BAD_ARG;
  if (!hptr) return IPA_OK;
  IPA_HIP(c, hipHostFree(hptr));
  return IPA_OK;
}

int ipa_memcpy_h2d(ipa_ctx* c, void* d, const void* h, size_t bytes) {
  if (!c || (bytes && (!d || !h))) return IPA_ERR_BAD_ARG;
  if (!bytes) return IPA_OK;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  IPA_HIP(c, hipStreamSynchronize(c->stream));
  return IPA_OK;
}

int ipa_memcpy_d2h(ipa_ctx* c, void* h, const void* d, size_t bytes) {
  if (!c || (bytes && (!d || !h))) return IPA_ERR_BAD_ARG;
  if (!bytes) return IPA_OK;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  IPA_HIP(c, hipStreamSynchronize(c->stream));
  return IPA_OK;
}

int ipa_memcpy_d2d(ipa_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || (bytes && (!dst || !src))) return IPA_ERR_BAD_ARG;
  if (!bytes) return IPA_OK;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return IPA_OK;
}

int ipa_memset(ipa_ctx* c, void* d, int value, size_t bytes) {
  if (!c || (bytes && !d)) return IPA_ERR_BAD_ARG;
  if (!bytes) return IPA_OK;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipMemsetAsync(d, value, bytes, c->stream));
  return IPA_OK;
}

int ipa_event_create(ipa_ctx* c, ipa_event** ev) {
  if (!c || !ev) return IPA_ERR_BAD_ARG;
  IPA_HIP(c, hipSetDevice(c->device));
  ipa_event* e = new ipa_event();
  hipError_t r = hipEventCreate(&e->ev);
  if (r != hipSuccess) {
    delete e;
    ipa_set_error(c, "hipEventCreate: %s", hipGetErrorString(r));
    return IPA_ERR_HIP;
  }
  *ev = e;
  return IPA_OK;
}

int ipa_event_destroy(ipa_ctx* c, ipa_event* ev) {
  if (!c) return IPA_ERR_BAD_ARG;
  if (!ev) return IPA_OK;
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return IPA_OK;
}

int ipa_event_record(ipa_ctx* c, ipa_event* ev) {
  if (!c || !ev) return IPA_ERR_BAD_ARG;
  IPA_HIP(c, hipEventRecord(ev->ev, c->stream));
  return IPA_OK;
}

int ipa_event_elapsed_ms(ipa_ctx* c, ipa_event* a, ipa_event* b, float* ms) {
  if (!c || !a || !b || !ms) return IPA_ERR_BAD_ARG;
  IPA_HIP(c, hipEventSynchronize(b->ev));
  IPA_HIP(c, hipEventElapsedTime(ms, a->ev, b->ev));
  return IPA_OK;
}

}  // extern "C"

int ipa_ws_reserve(ipa_ctx* c, size_t bytes) {
  if (c->ws_bytes >= bytes) return IPA_OK;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipStreamSynchronize(c->stream));
  if (c->ws) {
    IPA_HIP(c, hipFree(c->ws));
    c->ws = nullptr;
    c->ws_bytes = 0;
  }
  size_t want = bytes + (bytes >> 2) + (1u << 20);
  IPA_HIP(c, hipMalloc(&c->ws, want));
  c->ws_bytes = want;
  return IPA_OK;
}

int ipa_plan_reserve(ipa_ctx* c, size_t bytes) {
  c->plan_key_n = 0;  // the caller overwrites the buffer
  if (c->plan_bytes >= bytes) return IPA_OK;
  IPA_HIP(c, hipSetDevice(c->device));
  IPA_HIP(c, hipStreamSynchronize(c->stream));
  if (c->plan) {
    IPA_HIP(c, hipFree(c->plan));
    c->plan = nullptr;
    c->plan_bytes = 0;
  }
  size_t want = bytes + (bytes >> 1) + (1u << 16);
  IPA_HIP(c, hipMalloc(&c->plan, want));
  c->plan_bytes = want;
  return IPA_OK;
}

// Upload a small host table through pinned staging.  The previous user of the
// arena may still be running, and the pinned buffer may still be read by an
// earlier async copy, so drain the stream first: tables are only used by the
// secondary entry points (IDW, Lanczos), never by the per-frame hot loop.
int ipa_tab_upload(ipa_ctx* c, const void* host, size_t bytes, void** d) {
  IPA_HIP(c, hipSetDevice(c->device));
  // the same table as last time (resizes of one shape over the frames of a sequence): nothing to
  // send, and no host synchronisation between the calls
  if (c->tab && bytes && c->tab_valid == bytes && memcmp(c->tab_pinned, host, bytes) == 0) {
    *d = c->tab;
    return IPA_OK;
  }
  c->tab_valid = 0;
  if (c->tab_bytes < bytes) {
    IPA_HIP(c, hipStreamSynchronize(c->stream));
    if (c->tab) IPA_HIP(c, hipFree(c->tab));
    if (c->tab_pinned) IPA_HIP(c, hipHostFree(c->tab_pinned));
    c->tab = nullptr;
    c->tab_pinned = nullptr;
    c->tab_bytes = 0;
    size_t want = bytes < (64u << 10) ? (64u << 10) : bytes * 2;
    IPA_HIP(c, hipMalloc(&c->tab, want));
    IPA_HIP(c, hipHostMalloc(&c->tab_pinned, want, hipHostMallocDefault));
    c->tab_bytes = want;
  }
  IPA_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(c->tab_pinned, host, bytes);
  IPA_HIP(c, hipMemcpyAsync(c->tab, c->tab_pinned, bytes, hipMemcpyHostToDevice, c->stream));
  c->tab_valid = bytes;
  c->tab_serial++;
  *d = c->tab;
  return IPA_OK;
}
