// swt_lower.hip -- `str.lower()` on the device for the code points it maps one-to-one at equal UTF-8 length
// (the reference lowercases on the host at /root/reference/source/utils.py:27 -- via
// SubwordTokenizer.preprocessing -- and /root/reference/source/wordpiece.py:248).
//
// One thread per byte; the thread of a lead byte decodes its code point, looks the lowercase up in a dense table and, when
// it differs, re-encodes it over the same bytes (same length by construction of the table, tools/gen_lower_table.py).
// A sentence that holds one of the 26 code points whose lowercase has another length, several code points, or depends on
// its neighbours (U+03A3) is FLAGGED and left to the host: the caller lowercases that sentence itself.
#include <mutex>
#include <vector>

#include "swt_common.h"
#include "unicode_lower.inc"

namespace swt {

constexpr uint32_t kLowerHost = 0xFFFFFFFFu;

static std::vector<uint32_t> g_lower;
static std::once_flag g_lower_once;
static uint32_t *g_lower_dev = nullptr;
static std::mutex g_lower_mu;

static const uint32_t *host_lower_table() {
  std::call_once(g_lower_once, [] {
    g_lower.resize(kNumCodePoints);
    for (uint32_t c = 0; c < kNumCodePoints; c++) g_lower[c] = c;
    for (unsigned int i = 0; i < SWT_LOWER_NPAIRS; i++) g_lower[SWT_LOWER_PAIRS[i][0]] = SWT_LOWER_PAIRS[i][1];
    for (unsigned int i = 0; i < SWT_LOWER_NHOST; i++) g_lower[SWT_LOWER_HOST[i]] = kLowerHost;
  });
  return g_lower.data();
}

static int device_lower_table(const uint32_t **d) {
  std::lock_guard<std::mutex> lk(g_lower_mu);
  if (!g_lower_dev) {
    const uint32_t *h = host_lower_table();
    SWT_HIP(hipMalloc((void **)&g_lower_dev, (size_t)kNumCodePoints * 4));
    SWT_HIP(hipMemcpy(g_lower_dev, h, (size_t)kNumCodePoints * 4, hipMemcpyHostToDevice));
  }
  *d = g_lower_dev;
  return SWT_OK;
}

__global__ __launch_bounds__(256) void lower_kernel(uint8_t *__restrict__ text, uint64_t n_bytes, const uint64_t *__restrict__ sent_off,
                                                    uint64_t n_sent, const uint32_t *__restrict__ lower, uint8_t *__restrict__ need_host) {
  for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_bytes; g += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t b = text[g];
    if (b < 0x80u) {
      if (b - 'A' < 26u) text[g] = (uint8_t)(b + 32u);
      continue;
    }
    if (b < 0xC0u) continue;  // continuation bytes are rewritten by their lead byte's thread
    uint32_t len = (uint32_t)utf8_len((uint8_t)b);
    if (len < 2 || g + len > n_bytes) continue;  // 0xF8..: passed through; a truncated tail is left alone
    uint32_t cp = b & (0xFFu >> (len + 1));
    for (uint32_t i = 1; i < len; i++) cp = (cp << 6) | (text[g + i] & 0x3Fu);
    if (cp >= kNumCodePoints) continue;
    const uint32_t lo = lower[cp];
    if (lo == cp) continue;
    if (lo == kLowerHost) {
      // the sentence that holds byte g: the last one whose offset is <= g
      uint64_t a = 0, z = n_sent;
      while (a < z) {
        const uint64_t mid = (a + z) >> 1;
        if (sent_off[mid + 1] <= g) a = mid + 1; else z = mid;
      }
      if (a < n_sent) need_host[a] = 1;
      continue;
    }
    // same UTF-8 length by construction
    if (len == 2) {
      text[g] = (uint8_t)(0xC0u | (lo >> 6));
      text[g + 1] = (uint8_t)(0x80u | (lo & 0x3Fu));
    } else if (len == 3) {
      text[g] = (uint8_t)(0xE0u | (lo >> 12));
      text[g + 1] = (uint8_t)(0x80u | ((lo >> 6) & 0x3Fu));
      text[g + 2] = (uint8_t)(0x80u | (lo & 0x3Fu));
    } else {
      text[g] = (uint8_t)(0xF0u | (lo >> 18));
      text[g + 1] = (uint8_t)(0x80u | ((lo >> 12) & 0x3Fu));
      text[g + 2] = (uint8_t)(0x80u | ((lo >> 6) & 0x3Fu));
      text[g + 3] = (uint8_t)(0x80u | (lo & 0x3Fu));
    }
  }
}

}  // namespace swt

using namespace swt;

extern "C" {

uint32_t swt_lower_of(uint32_t cp) { return cp < kNumCodePoints ? host_lower_table()[cp] : cp; }

int swt_utf8_lower_dev(uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent, uint8_t *d_need_host,
                       void *stream) {
  if (!d_sent_off || (n_bytes && !d_text) || (n_sent && !d_need_host)) return fail(SWT_ERR_INVALID, "null argument");
  int rc = ensure_device();
  if (rc) return rc;
  const uint32_t *d_lower = nullptr;
  if ((rc = device_lower_table(&d_lower))) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (n_sent) SWT_HIP(hipMemsetAsync(d_need_host, 0, n_sent, st));
  if (n_bytes) {
    uint64_t blocks = (n_bytes + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(lower_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_text, n_bytes, d_sent_off, n_sent, d_lower, d_need_host);
  }
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

int swt_utf8_lower(uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, uint8_t *need_host) {
  if (!sent_off || (n_sent && !need_host)) return fail(SWT_ERR_INVALID, "null argument");
  if (sent_off[0] != 0) return fail(SWT_ERR_INVALID, "sent_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (sent_off[s] > sent_off[s + 1]) return fail(SWT_ERR_INVALID, "sentence offsets must be non-decreasing");
  const uint64_t n_bytes = sent_off[n_sent];
  if (n_bytes && !text) return fail(SWT_ERR_INVALID, "null text");
  int rc = ensure_device();
  if (rc) return rc;
  DevBuf d_text, d_off, d_flag;
  if ((rc = d_text.reserve(n_bytes + 16)) || (rc = d_off.reserve((n_sent + 1) * 8)) || (rc = d_flag.reserve(n_sent + 16))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpy(d_text.p, text, n_bytes, hipMemcpyHostToDevice));
  SWT_HIP(hipMemcpy(d_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice));
  rc = swt_utf8_lower_dev(d_text.as<uint8_t>(), n_bytes, d_off.as<uint64_t>(), n_sent, d_flag.as<uint8_t>(), nullptr);
  if (!rc && n_bytes) SWT_HIP(hipMemcpy(text, d_text.p, n_bytes, hipMemcpyDeviceToHost));
  if (!rc && n_sent) SWT_HIP(hipMemcpy(need_host, d_flag.p, n_sent, hipMemcpyDeviceToHost));
  d_text.release(); d_off.release(); d_flag.release();
  return rc;
}

}  // extern "C"
