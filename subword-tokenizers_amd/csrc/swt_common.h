// swt_common.h -- shared host/device helpers of libswt_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/swt.h"

namespace swt {

// ---- errors ------------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define SWT_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return ::swt::fail(SWT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// The exception barrier of the C ABI (include/swt.h: "nothing throws, aborts"): every extern "C" entry point is a
// function-try-block that ends in SWT_API_CATCH, so a std::bad_alloc / std::length_error from a host-side container (or
// anything else) becomes SWT_ERR_NOMEM / SWT_ERR_INTERNAL + swt_last_error() instead of std::terminate -> SIGABRT.
// api_exception() must be called inside a catch block (it rethrows to classify).
int api_exception() noexcept;
#define SWT_API_CATCH catch (...) { return ::swt::api_exception(); }
#define SWT_API_CATCH_VOID catch (...) { (void)::swt::api_exception(); }

int ensure_device();  // SWT_OK when a device is selected (selects 0 on first use)
int device_cus();

// Ablation switches of the encode kernels (skip a phase: results are WRONG while set).  They exist only in a library built
// with -DSWT_ABLATION (tools/gpu_ablate*.py); the product build folds them to zero.
#ifdef SWT_ABLATION
int ablation_knob(int which);
#else
constexpr int ablation_knob(int) { return 0; }
#endif

// ---- dominant-kernel timing (bench.py roofline) ---------------------------------------------------
// level 1 = the dominant kernel of a path, 2 = all kernels of a call, 3.. = diagnostics (swt_profile_enable selects one)
void prof_begin(hipStream_t st, int level = 1);
void prof_end(hipStream_t st, int level = 1);

// ---- device buffers ----------------------------------------------------------------------------
// Grow-only device buffer (reused across calls so the hot path never allocates in steady state).
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);
  void release();
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// pinned host staging (grow-only): one async copy each way for the small calls of the reference-style entry points
struct PinnedBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes);
  void release();
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};
constexpr uint64_t kSmallCallBytes = 64 * 1024;  // text bytes up to which the host entry points take the one-copy path
constexpr uint64_t kSmallCallSents = 4096;

// ---- class table -------------------------------------------------------------------------------
constexpr uint32_t kNumCodePoints = 0x110000u;
const uint8_t *host_class_table();          // 0x110000 bytes, bits SWT_CLS_*
int device_class_table(const uint8_t **d);  // uploaded once per process

// ---- hashing (host and device must agree) --------------------------------------------------------
constexpr uint64_t kEmptyKey = ~0ull;
__host__ __device__ inline uint32_t hash_slot(uint64_t key, uint32_t bits) {
  return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> (64 - bits));
}
__host__ __device__ inline uint64_t pair_key(uint32_t l, uint32_t r) { return ((uint64_t)l << 32) | r; }

// ---- UTF-8 ---------------------------------------------------------------------------------------
// Generalised UTF-8 (surrogates pass through).  Length from the lead byte; stray continuation bytes
// and 0xF8.. count as one byte.  Never reads outside [p, end).
__host__ __device__ inline int utf8_len(uint8_t b) {
  return b < 0x80 ? 1 : (b < 0xC0 ? 1 : (b < 0xE0 ? 2 : (b < 0xF0 ? 3 : (b < 0xF8 ? 4 : 1))));
}
__host__ __device__ inline bool utf8_is_cont(uint8_t b) { return (b & 0xC0) == 0x80; }

inline uint32_t utf8_decode_host(const uint8_t *p, const uint8_t *end, int *len) {
  uint8_t b = p[0];
  int n = utf8_len(b);
  if (p + n > end) n = (int)(end - p);
  *len = n < 1 ? 1 : n;
  if (b < 0x80 || n <= 1) return b;
  uint32_t cp = b & (0xFF >> (n + 1));
  for (int i = 1; i < n; i++) cp = (cp << 6) | (p[i] & 0x3F);
  return cp;
}

constexpr uint32_t kInvalidTok = 0xFFFFFFFFu;

}  // namespace swt
