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
#include "swt_tile.h"
#include "swt_words.h"
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

// ---- sentence offsets in BYTES from offsets in CODE POINTS ---------------------------------------------------------------
// The host can join its strings and encode them in one call, and it knows every string's length in code points for free
// (len(str)); what costs it 20 ms per 8.5 MB is the length of every string in BYTES.  A code point is a byte that is not a
// UTF-8 continuation byte, so the device counts those per 1-KiB block, scans the counts, and one wave per sentence finds
// the byte at which its first code point starts.
constexpr uint32_t kOffBlock = 1024;

__global__ __launch_bounds__(64) void lead_count_kernel(const uint8_t *__restrict__ text, uint64_t n_bytes, uint32_t *__restrict__ blk_cnt) {
  const uint64_t b0 = (uint64_t)blockIdx.x * kOffBlock;
  uint32_t c = 0;
  for (uint32_t i = threadIdx.x * 16; i < kOffBlock; i += 64 * 16) {
    const uint64_t g = b0 + i;
    if (g + 16 <= n_bytes && ((reinterpret_cast<uintptr_t>(text + g) & 15) == 0)) {
      const uint4 v = *reinterpret_cast<const uint4 *>(text + g);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        // continuation bytes are 10xxxxxx: bit 7 set and bit 6 clear
        const uint32_t cont = (w[k] >> 7) & ~(w[k] >> 6) & 0x01010101u;
        c += 4u - (uint32_t)__popc(cont);
      }
    } else {
      for (int k = 0; k < 16; k++)
        if (g + k < n_bytes && (text[g + k] & 0xC0u) != 0x80u) c++;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = c;
}

__global__ __launch_bounds__(64) void cp_to_byte_kernel(const uint8_t *__restrict__ text, uint64_t n_bytes, const uint64_t *__restrict__ cp_off,
                                                        uint64_t n_off, uint64_t n_blocks, const uint32_t *__restrict__ blk_local,
                                                        const unsigned long long *__restrict__ blk_base, uint64_t *__restrict__ byte_off) {
  const uint64_t s = blockIdx.x;
  if (s >= n_off) return;
  const int lane = threadIdx.x;
  const uint64_t k = cp_off[s];  // code points before this sentence
  // the block that holds code point k: the last one whose exclusive prefix is <= k
  uint64_t a = 0, z = n_blocks;
  while (a + 1 < z) {
    const uint64_t mid = (a + z) >> 1;
    if (blk_base[mid >> 10] + blk_local[mid] <= k) a = mid; else z = mid;
  }
  uint64_t need = k - (blk_base[a >> 10] + blk_local[a]);  // leads to pass inside block a
  uint64_t pos = a * (uint64_t)kOffBlock;
  uint64_t found = n_bytes;  // k == total code points: the end of the text
  for (uint32_t i = 0; i < kOffBlock && pos + i < n_bytes; i += 64) {
    const uint64_t g = pos + i + lane;
    const bool lead = g < n_bytes && (text[g] & 0xC0u) != 0x80u;
    const unsigned long long M = __ballot(lead);
    const uint32_t c = (uint32_t)__popcll(M);
    if (need < c) {
      // the (need)-th set bit of M
      unsigned long long m = M;
      for (uint64_t j = 0; j < need; j++) m &= m - 1ull;
      found = pos + i + (uint64_t)__builtin_ctzll(m);
      break;
    }
    need -= c;
  }
  if (lane == 0) byte_off[s] = found;
}

// ---- sentence offsets from a SEPARATOR -----------------------------------------------------------------------------------
// Cheaper still for the host: join the strings with U+0000 between them (one call, like the plain join), let the device find
// the separators, close the gaps and note where every sentence starts.  (The caller has made sure that the text holds no
// U+0000 of its own: exactly n_sent - 1 zero bytes.)  Same block scheme as above: zero bytes per 1-KiB block, a scan, and
// a pass that moves every other byte down by the number of separators before it.
__global__ __launch_bounds__(64) void sep_count_kernel(const uint8_t *__restrict__ text, uint64_t n_bytes, uint32_t *__restrict__ blk_cnt) {
  const uint64_t b0 = (uint64_t)blockIdx.x * kOffBlock;
  uint32_t c = 0;
  for (uint32_t i = threadIdx.x; i < kOffBlock; i += 64) {
    const uint64_t g = b0 + i;
    if (g < n_bytes && text[g] == 0) c++;
  }
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = c;
}

// `out` holds n_bytes - (n_sent - 1) bytes: right when the text holds exactly n_sent - 1 separators.  The host only learns the
// true count from the scan's total, which it reads AFTER this launch (one synchronisation instead of two), so every write is
// bounded by what was allocated: with fewer separators than announced the tail of the text is dropped here -- not written
// past the buffer -- and the host then rejects the call (SWT_ERR_INVALID).
__global__ __launch_bounds__(64) void sep_split_kernel(const uint8_t *__restrict__ text, uint64_t n_bytes, uint64_t n_sent,
                                                       const uint32_t *__restrict__ blk_local, const unsigned long long *__restrict__ blk_base,
                                                       uint8_t *__restrict__ out, uint64_t *__restrict__ byte_off) {
  const uint64_t out_cap = n_bytes - (n_sent ? n_sent - 1 : 0);
  const uint64_t b = blockIdx.x, b0 = b * kOffBlock;
  const int lane = threadIdx.x;
  uint64_t z = blk_base[b >> 10] + blk_local[b];  // separators before this block
  if (b == 0 && lane == 0) {
    byte_off[0] = 0;
    byte_off[n_sent] = n_bytes - (n_sent ? n_sent - 1 : 0);
  }
  for (uint32_t i = 0; i < kOffBlock && b0 + i < n_bytes; i += 64) {
    const uint64_t g = b0 + i + lane;
    const uint8_t c = g < n_bytes ? text[g] : (uint8_t)1;
    const bool sep = g < n_bytes && c == 0;
    const unsigned long long M = __ballot(sep);
    const uint64_t mine = z + (uint64_t)__popcll(M & ((1ull << lane) - 1ull));  // separators before this byte
    if (g < n_bytes) {
      if (!sep) { if (g - mine < out_cap) out[g - mine] = c; }
      else if (mine + 1 < n_sent) byte_off[mine + 1] = g - mine;  // sentence mine + 1 starts behind this separator
    }
    z += (uint64_t)__popcll(M);
  }
}

}  // namespace swt

using namespace swt;

// The device buffers of the host entry points below: grow-only and kept between calls of the calling thread (a call used to
// pay four or five hipMalloc / hipFree pairs), given back when a call ends holding more than kPrepareKeep bytes, and released
// on every error path by the guard that owns them.
namespace {
constexpr size_t kPrepareKeep = (size_t)256 << 20;
struct PrepareWs {
  DevBuf in, text, cp, off, flag;
  TileWorkspace ws;
  void release() { in.release(); text.release(); cp.release(); off.release(); flag.release(); ws.release(); }
  size_t held() const { return in.cap + text.cap + cp.cap + off.cap + flag.cap; }
  ~PrepareWs() { release(); }
};
struct PrepareGuard {  // an error in mid call: nothing of a half-written state is kept
  PrepareWs &w;
  bool ok = false;
  ~PrepareGuard() { if (!ok || w.held() > kPrepareKeep) w.release(); }
};
PrepareWs &prepare_ws() {
  static thread_local PrepareWs w;
  return w;
}
}  // namespace

namespace swt {
// The separator form on the device: the text without its separators, lowercased, and the sentence offsets stay in the calling
// thread's workspace (valid until its next prepare / lower call) for a consumer on the device (swt_bpe_train_create_joined);
// need_host[] comes back to the host.  The caller has checked the arguments.
int prepare_joined_dev(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, const uint8_t **d_text,
                       const uint64_t **d_off, uint64_t *n_bytes_out) {
  if (n_sent == 0 ? n_joined != 0 : n_joined + 1 < n_sent) return fail(SWT_ERR_INVALID, "fewer bytes than separators");
  if (n_sent + 1 > 0x7FFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "too many sentences for one call");
  int rc = ensure_device();
  if (rc) return rc;
  PrepareWs &W = prepare_ws();
  const uint64_t n_bytes = n_joined - (n_sent ? n_sent - 1 : 0);
  const uint64_t n_blocks = n_joined ? (n_joined + kOffBlock - 1) / kOffBlock : 1;
  if ((rc = W.in.reserve(n_joined + 16)) || (rc = W.text.reserve(n_bytes + 64)) || (rc = W.off.reserve((n_sent + 1) * 8)) ||
      (rc = W.flag.reserve(n_sent + 16)) || (rc = W.ws.reserve(0, 0, n_blocks)))
    return rc;
  if (n_joined) SWT_HIP(hipMemcpy(W.in.p, joined, n_joined, hipMemcpyHostToDevice));
  SWT_HIP(hipMemset(W.off.p, 0xFF, (n_sent + 1) * 8));  // a start that no separator announces stays ~0: the caller counted wrong
  hipLaunchKernelGGL(sep_count_kernel, dim3((unsigned)n_blocks), dim3(64), 0, nullptr, W.in.as<uint8_t>(), n_joined, W.ws.tile_tok.as<uint32_t>());
  launch_scan_only(n_blocks, W.ws, W.ws.plan.as<uint64_t>(), nullptr);
  const uint64_t nb = (n_blocks + 1023) / 1024;
  hipLaunchKernelGGL(sep_split_kernel, dim3((unsigned)n_blocks), dim3(64), 0, nullptr, W.in.as<uint8_t>(), n_joined, n_sent,
                     W.ws.tile_base.as<uint32_t>(), W.ws.blk.as<unsigned long long>() + 1 + nb, W.text.as<uint8_t>(), W.off.as<uint64_t>());
  SWT_HIP(hipGetLastError());
  uint64_t n_sep = 0;  // the scan's total: with exactly n_sent - 1 separators every sentence start was written once
  SWT_HIP(hipMemcpy(&n_sep, W.ws.plan.p, 8, hipMemcpyDeviceToHost));
  if (n_sep != (n_sent ? n_sent - 1 : 0)) return fail(SWT_ERR_INVALID, "the joined text holds %llu separators, not n_sent - 1", (unsigned long long)n_sep);
  if ((rc = swt_utf8_lower_dev(W.text.as<uint8_t>(), n_bytes, W.off.as<uint64_t>(), n_sent, W.flag.as<uint8_t>(), nullptr))) return rc;
  if (n_sent) SWT_HIP(hipMemcpy(need_host, W.flag.p, n_sent, hipMemcpyDeviceToHost));
  if (d_text) *d_text = W.text.as<uint8_t>();
  if (d_off) *d_off = W.off.as<uint64_t>();
  if (n_bytes_out) *n_bytes_out = n_bytes;
  return SWT_OK;
}

int with_prepared_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *need_host, bool *consumed,
                         PreparedConsumer consume, void *ctx) {
  *consumed = false;
  PrepareWs &W = prepare_ws();
  PrepareGuard guard{W};
  const uint8_t *d_text = nullptr;
  const uint64_t *d_off = nullptr;
  uint64_t n_bytes = 0;
  int rc = prepare_joined_dev(joined, n_joined, n_sent, need_host, &d_text, &d_off, &n_bytes);
  if (rc) return rc;
  bool host = false;
  for (uint64_t s2 = 0; s2 < n_sent && !host; s2++) host = need_host[s2] != 0;
  if (!host) {
    if ((rc = consume(ctx, d_text, n_bytes, d_off))) return rc;
    *consumed = true;
  }
  guard.ok = true;
  return SWT_OK;
}
}  // namespace swt

extern "C" {

uint32_t swt_lower_of(uint32_t cp) try { return cp < kNumCodePoints ? host_lower_table()[cp] : cp; } catch (...) { (void)::swt::api_exception(); return 0; }

const char *swt_unidata_version(void) { return SWT_UNIDATA_VERSION; }

int swt_utf8_lower_dev(uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_sent_off, uint64_t n_sent, uint8_t *d_need_host,
                       void *stream) try {
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
} SWT_API_CATCH

// d_byte_off[s] = byte at which the sentence that starts after d_cp_off[s] code points begins (n_sent + 1 entries; well-formed
// UTF-8: a code point is a byte that is not a continuation byte)
static int utf8_offsets_dev(const uint8_t *d_text, uint64_t n_bytes, const uint64_t *d_cp_off, uint64_t n_sent, uint64_t *d_byte_off,
                            TileWorkspace &ws, hipStream_t st) {
  const uint64_t n_blocks = n_bytes ? (n_bytes + kOffBlock - 1) / kOffBlock : 1;
  int rc;
  if ((rc = ws.reserve(0, 0, n_blocks))) return rc;  // tile_tok = counts, tile_base / blk = their scan, plan[0] = the total
  hipLaunchKernelGGL(lead_count_kernel, dim3((unsigned)n_blocks), dim3(64), 0, st, d_text, n_bytes, ws.tile_tok.as<uint32_t>());
  launch_scan_only(n_blocks, ws, ws.plan.as<uint64_t>(), st);
  const uint64_t nb = (n_blocks + 1023) / 1024;
  hipLaunchKernelGGL(cp_to_byte_kernel, dim3((unsigned)(n_sent + 1)), dim3(64), 0, st, d_text, n_bytes, d_cp_off, n_sent + 1, n_blocks,
                     ws.tile_base.as<uint32_t>(), ws.blk.as<unsigned long long>() + 1 + nb, d_byte_off);
  SWT_HIP(hipGetLastError());
  return SWT_OK;
}

int swt_utf8_prepare(uint8_t *text, uint64_t n_bytes, const uint64_t *cp_off, uint64_t n_sent, uint64_t *byte_off, uint8_t *need_host) try {
  if (!cp_off || !byte_off || (n_sent && !need_host) || (n_bytes && !text)) return fail(SWT_ERR_INVALID, "null argument");
  if (cp_off[0] != 0) return fail(SWT_ERR_INVALID, "cp_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (cp_off[s] > cp_off[s + 1]) return fail(SWT_ERR_INVALID, "offsets must be non-decreasing");
  if (cp_off[n_sent] > n_bytes) return fail(SWT_ERR_INVALID, "more code points than bytes");
  if (n_sent + 1 > 0x7FFFFFFFull) return fail(SWT_ERR_UNSUPPORTED, "too many sentences for one call");
  int rc = ensure_device();
  if (rc) return rc;
  PrepareWs &W = prepare_ws();
  PrepareGuard guard{W};
  DevBuf &d_text = W.text, &d_cp = W.cp, &d_off = W.off, &d_flag = W.flag;
  TileWorkspace &ws = W.ws;
  if ((rc = d_text.reserve(n_bytes + 16)) || (rc = d_cp.reserve((n_sent + 1) * 8)) || (rc = d_off.reserve((n_sent + 1) * 8)) ||
      (rc = d_flag.reserve(n_sent + 16)))
    return rc;
  if (n_bytes) SWT_HIP(hipMemcpy(d_text.p, text, n_bytes, hipMemcpyHostToDevice));
  SWT_HIP(hipMemcpy(d_cp.p, cp_off, (n_sent + 1) * 8, hipMemcpyHostToDevice));
  rc = utf8_offsets_dev(d_text.as<uint8_t>(), n_bytes, d_cp.as<uint64_t>(), n_sent, d_off.as<uint64_t>(), ws, nullptr);
  if (!rc) rc = swt_utf8_lower_dev(d_text.as<uint8_t>(), n_bytes, d_off.as<uint64_t>(), n_sent, d_flag.as<uint8_t>(), nullptr);
  if (!rc) {
    SWT_HIP(hipMemcpy(byte_off, d_off.p, (n_sent + 1) * 8, hipMemcpyDeviceToHost));
    if (n_bytes) SWT_HIP(hipMemcpy(text, d_text.p, n_bytes, hipMemcpyDeviceToHost));
    if (n_sent) SWT_HIP(hipMemcpy(need_host, d_flag.p, n_sent, hipMemcpyDeviceToHost));
    if (byte_off[n_sent] != n_bytes)
      rc = fail(SWT_ERR_INVALID, "cp_off[n_sent] = %llu is not the number of code points in the text", (unsigned long long)cp_off[n_sent]);
  }
  guard.ok = rc == SWT_OK;
  return rc;
} SWT_API_CATCH

int swt_utf8_prepare_joined(const uint8_t *joined, uint64_t n_joined, uint64_t n_sent, uint8_t *text_out, uint64_t *byte_off, uint8_t *need_host) try {
  if (!byte_off || (n_sent && !need_host) || (n_joined && (!joined || !text_out))) return fail(SWT_ERR_INVALID, "null argument");
  PrepareWs &W = prepare_ws();
  PrepareGuard guard{W};
  uint64_t n_bytes = 0;
  int rc = swt::prepare_joined_dev(joined, n_joined, n_sent, need_host, nullptr, nullptr, &n_bytes);
  if (!rc) {
    SWT_HIP(hipMemcpy(byte_off, W.off.p, (n_sent + 1) * 8, hipMemcpyDeviceToHost));
    if (n_bytes) SWT_HIP(hipMemcpy(text_out, W.text.p, n_bytes, hipMemcpyDeviceToHost));
  }
  guard.ok = rc == SWT_OK;
  return rc;
} SWT_API_CATCH

int swt_utf8_lower(uint8_t *text, const uint64_t *sent_off, uint64_t n_sent, uint8_t *need_host) try {
  if (!sent_off || (n_sent && !need_host)) return fail(SWT_ERR_INVALID, "null argument");
  if (sent_off[0] != 0) return fail(SWT_ERR_INVALID, "sent_off[0] must be 0");
  for (uint64_t s = 0; s < n_sent; s++)
    if (sent_off[s] > sent_off[s + 1]) return fail(SWT_ERR_INVALID, "sentence offsets must be non-decreasing");
  const uint64_t n_bytes = sent_off[n_sent];
  if (n_bytes && !text) return fail(SWT_ERR_INVALID, "null text");
  int rc = ensure_device();
  if (rc) return rc;
  PrepareWs &W = prepare_ws();
  PrepareGuard guard{W};
  DevBuf &d_text = W.text, &d_off = W.off, &d_flag = W.flag;
  if ((rc = d_text.reserve(n_bytes + 16)) || (rc = d_off.reserve((n_sent + 1) * 8)) || (rc = d_flag.reserve(n_sent + 16))) return rc;
  if (n_bytes) SWT_HIP(hipMemcpy(d_text.p, text, n_bytes, hipMemcpyHostToDevice));
  SWT_HIP(hipMemcpy(d_off.p, sent_off, (n_sent + 1) * 8, hipMemcpyHostToDevice));
  rc = swt_utf8_lower_dev(d_text.as<uint8_t>(), n_bytes, d_off.as<uint64_t>(), n_sent, d_flag.as<uint8_t>(), nullptr);
  if (!rc && n_bytes) SWT_HIP(hipMemcpy(text, d_text.p, n_bytes, hipMemcpyDeviceToHost));
  if (!rc && n_sent) SWT_HIP(hipMemcpy(need_host, d_flag.p, n_sent, hipMemcpyDeviceToHost));
  guard.ok = rc == SWT_OK;
  return rc;
} SWT_API_CATCH

}  // extern "C"
