// swt_core.hip -- process-level state of libswt_hip.so: error text, device selection, class tables.
#include <mutex>
#include <new>
#include <stdexcept>

#include "swt_common.h"
#include "unicode_classes.inc"

namespace swt {

static thread_local std::string g_err;

// (the assignment may allocate: an error path must not throw on its own)
static void keep_error(const char *text) noexcept {
  try { g_err = text; } catch (...) { }
}

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  keep_error(buf);
}

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  keep_error(buf);
  return code;
}

int api_exception() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc &) {
    return fail(SWT_ERR_NOMEM, "out of host memory (std::bad_alloc) inside libswt_hip");
  } catch (const std::length_error &e) {
    return fail(SWT_ERR_NOMEM, "a host container was asked for an impossible size (std::length_error: %s)", e.what());
  } catch (const std::exception &e) {
    return fail(SWT_ERR_INTERNAL, "C++ exception stopped at the ABI boundary: %s", e.what());
  } catch (...) {
    return fail(SWT_ERR_INTERNAL, "unknown C++ exception stopped at the ABI boundary");
  }
}

static int g_device = -1;
static int g_cus = 0;
static std::mutex g_mu;

static int select_device(int ordinal) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(SWT_ERR_NO_DEVICE, "no HIP device available (%s); libswt_hip has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (ordinal < 0 || ordinal >= n) return fail(SWT_ERR_INVALID, "device ordinal %d out of range [0,%d)", ordinal, n);
  SWT_HIP(hipSetDevice(ordinal));
  hipDeviceProp_t prop;
  SWT_HIP(hipGetDeviceProperties(&prop, ordinal));
  g_device = ordinal;
  g_cus = prop.multiProcessorCount;
  return SWT_OK;
}

int ensure_device() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_device >= 0) {
    // keep the calling thread on the selected device
    hipError_t e = hipSetDevice(g_device);
    if (e != hipSuccess) return fail(SWT_ERR_HIP, "hipSetDevice(%d): %s", g_device, hipGetErrorString(e));
    return SWT_OK;
  }
  return select_device(0);
}

int device_cus() { return g_cus > 0 ? g_cus : 256; }

int DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return SWT_OK;
  size_t want = bytes + bytes / 4 + 256;
  void *np = nullptr;
  SWT_HIP(hipMalloc(&np, want));
  if (p) (void)hipFree(p);
  p = np;
  cap = want;
  return SWT_OK;
}

int PinnedBuf::reserve(size_t bytes) {
  if (bytes <= cap) return SWT_OK;
  const size_t want = bytes + bytes / 2 + 4096;
  void *np = nullptr;
  SWT_HIP(hipHostMalloc(&np, want, hipHostMallocDefault));
  if (p) (void)hipHostFree(p);
  p = np;
  cap = want;
  return SWT_OK;
}

void PinnedBuf::release() {
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
}

void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

#ifdef SWT_ABLATION
static int g_knobs[8] = {0};
int ablation_knob(int which) { return (which >= 0 && which < 8) ? g_knobs[which] : 0; }
#endif

// timing state of the calling thread (a handle is used by one host thread at a time: nothing here is shared)
static thread_local int g_prof_on = 0;  // 0 off, else the level that is being timed
static thread_local std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pending, g_prof_pool;

void prof_begin(hipStream_t st, int level) {
  if (g_prof_on != level) return;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!g_prof_pool.empty()) {
    ev = g_prof_pool.back();
    g_prof_pool.pop_back();
  } else {
    if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return;
  }
  (void)hipEventRecord(ev.first, st);
  g_prof_pending.push_back(ev);
}

void prof_end(hipStream_t st, int level) {
  if (g_prof_on != level || g_prof_pending.empty()) return;
  (void)hipEventRecord(g_prof_pending.back().second, st);
}

static std::vector<uint8_t> g_cls;
static std::once_flag g_cls_once;
static uint8_t *g_cls_dev = nullptr;

static void fill(const unsigned int (*r)[2], unsigned int n, uint8_t bit) {
  for (unsigned int i = 0; i < n; i++)
    for (unsigned int c = r[i][0]; c <= r[i][1]; c++) g_cls[c] |= bit;
}

const uint8_t *host_class_table() {
  std::call_once(g_cls_once, [] {
    g_cls.assign(kNumCodePoints, 0);
    fill(SWT_BERT_WS_RANGES, SWT_BERT_WS_NRANGES, SWT_CLS_BERT_WS);
    fill(SWT_BERT_PUNCT_RANGES, SWT_BERT_PUNCT_NRANGES, SWT_CLS_BERT_PUNCT);
    fill(SWT_PY_SPACE_RANGES, SWT_PY_SPACE_NRANGES, SWT_CLS_PY_SPACE);
    fill(SWT_PY_ALNUM_RANGES, SWT_PY_ALNUM_NRANGES, SWT_CLS_PY_ALNUM);
  });
  return g_cls.data();
}

int device_class_table(const uint8_t **d) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_cls_dev) {
    const uint8_t *h = host_class_table();
    SWT_HIP(hipMalloc((void **)&g_cls_dev, kNumCodePoints));
    SWT_HIP(hipMemcpy(g_cls_dev, h, kNumCodePoints, hipMemcpyHostToDevice));
  }
  *d = g_cls_dev;
  return SWT_OK;
}

}  // namespace swt

extern "C" {

const char *swt_last_error(void) { return swt::g_err.c_str(); }

int swt_version(void) { return 2; }

int swt_abi_selftest(int kind) try {
  switch (kind) {
    case 0: return SWT_OK;
    case 1: throw std::bad_alloc();
    case 2: { std::vector<uint64_t> v; v.resize(v.max_size() + 1); return (int)v.size(); }
    case 3: throw std::runtime_error("swt_abi_selftest");
    case 4: throw 42;
    default: return swt::fail(SWT_ERR_INVALID, "no such self test");
  }
} SWT_API_CATCH

int swt_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} SWT_API_CATCH

int swt_init(int device_ordinal) try {
  std::lock_guard<std::mutex> lk(swt::g_mu);
  if (swt::g_device >= 0 && swt::g_device != device_ordinal && swt::g_cls_dev)
    return swt::fail(SWT_ERR_STATE, "device %d already selected for this process", swt::g_device);
  return swt::select_device(device_ordinal);
} SWT_API_CATCH

int swt_device_info(int *n_cu, char *name, size_t name_cap) try {
  int rc = swt::ensure_device();
  if (rc) return rc;
  hipDeviceProp_t prop;
  SWT_HIP(hipGetDeviceProperties(&prop, swt::g_device));
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (name && name_cap) {
    snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return SWT_OK;
} SWT_API_CATCH

#ifdef SWT_ABLATION
int swt_ablation_knob(int which, int value) {  // not declared in include/swt.h: ablation builds only
  if (which < 0 || which >= 8) return swt::fail(SWT_ERR_INVALID, "no such knob");
  swt::g_knobs[which] = value;
  return SWT_OK;
}
#endif

int swt_profile_enable(int on) try {
  swt::g_prof_on = on;
  return SWT_OK;
} SWT_API_CATCH

int swt_profile_read(double *ms_total, uint64_t *n_launches) try {
  double ms = 0;
  uint64_t n = 0;
  for (auto &ev : swt::g_prof_pending) {
    float t = 0;
    SWT_HIP(hipEventSynchronize(ev.second));
    SWT_HIP(hipEventElapsedTime(&t, ev.first, ev.second));
    ms += t;
    n++;
    swt::g_prof_pool.push_back(ev);
  }
  swt::g_prof_pending.clear();
  if (ms_total) *ms_total = ms;
  if (n_launches) *n_launches = n;
  return SWT_OK;
} SWT_API_CATCH

unsigned swt_class_of(uint32_t cp) try { return cp < swt::kNumCodePoints ? swt::host_class_table()[cp] : 0u; } catch (...) { (void)::swt::api_exception(); return 0; }

}  // extern "C"
