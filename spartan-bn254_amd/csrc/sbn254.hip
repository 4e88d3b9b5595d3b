// sbn254.hip — context, launch code and the C ABI (include/sbn254.h) of libsbn254_hip.so.
//
// Host-side mirror of the reference's operator boundary for the hot path:
//   GroupElement::msm_affine            src/group.rs:171-175        -> sbn_msm / sbn_msm_bases*
//   MultiCommitGens                     src/commitments.rs:17-114   -> sbn_bases_upload / sbn_gens_new
//   Commitments::commit, commit_inner   src/commitments.rs:144-154, src/hyrax.rs:253-308 -> sbn_commit_rows*
//   sumcheck prover loops / bind        src/sumcheck.rs, src/hyrax.rs:195-203            -> sbn_sc_* / sbn_bind_top
// There is no CPU fallback in this file: every entry point needs the gfx950 device.
#include "../../include/sbn254.h"
#include "host_field.hpp"
#include "msm_kernels.cuh"
#include "sumcheck_kernels.cuh"
#include "host_keccak.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

using namespace sbn;

// ------------------------------------------------------------------------------------------------
struct ProfEntry { std::string name; double ms = 0; uint64_t launches = 0; };
struct PendingEvt { int idx; hipEvent_t e0, e1; };

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
};

struct sbn_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::mutex mu;
  std::string err;
  // workspace (grown on demand, never shrunk; no allocation in steady state)
  DevBuf scal_canon, pts_mont, hist, offs, cursor, sorted, buckets, red_a, red_b, wsum, stage_scal, stage_pts, out_small;
  DevBuf sc_args, sc_partial, sc_out, sc_r, sc_tabs, gen_tmp, acc_ctr, extra_list, extra_out, big_list, digits, blockhist, size_bins, perm, merged;
  hipStream_t copy_stream = nullptr;          // H2D of the next row chunk while the current one is being committed
  hipEvent_t z_consumed = nullptr;            // set while a chunked commit is running: recorded when a chunk's scalars have been read
  DevBuf zstage[2], out_rows;
  bool sort_rows_ok = false;  // 160 KiB dynamic LDS granted to k_sort_rows
  int sort_rs_max = 16384;   // LDS counters per sort block (raised to 32768 when 128 KiB of dynamic LDS is granted)
  void* pin = nullptr; size_t pin_cap = 0;   // pinned host staging for small D2H results
  // profiling
  bool prof = false;
  std::vector<ProfEntry> prof_entries;
  std::vector<PendingEvt> prof_pending;
  std::vector<hipEvent_t> evt_pool;
};

struct sbn_bases {
  size_t n = 0;           // number of G points
  bool has_h = false;
  void* d_pts = nullptr;  // (n + has_h) x 64 B, Montgomery affine
  mutable std::unordered_map<int, void*> tables;   // window bits c -> W x (n + has_h) x 64 B: 2^(c w) * P_j (built on first commit)
  // equal bases merged (commit path): unique points as their own table + CSR of the columns that map to each
  sbn_bases* uniq = nullptr;
  size_t U = 0; uint32_t nbig = 0;
  void* d_csr_off = nullptr; void* d_csr_cols = nullptr; void* d_big = nullptr;
};
static const uint32_t MERGE_BIG = 64;
extern "C" void sbn_bases_free(sbn_ctx* c, sbn_bases* b);

struct sbn_table {
  void* d = nullptr; size_t len = 0; size_t cap = 0; bool owned = true;
  void* d2 = nullptr; size_t cap2 = 0; bool owned2 = true;     // second buffer for the fused (out-of-place) bind
};

static int fail(sbn_ctx* c, int code, const char* fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (c) c->err = buf;
  return code;
}
#define HIPCHK(c, call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail((c), SBN_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)

static int ensure(sbn_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return SBN_OK;
  if (b.p) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + (bytes >> 3);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { b.p = nullptr; return fail(c, SBN_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
  b.cap = want;
  return SBN_OK;
}
static int ensure_pin(sbn_ctx* c, size_t bytes) {
  if (bytes <= c->pin_cap) return SBN_OK;
  if (c->pin) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipHostFree(c->pin)); c->pin = nullptr; c->pin_cap = 0; }
  HIPCHK(c, hipHostMalloc(&c->pin, bytes, hipHostMallocDefault));
  c->pin_cap = bytes;
  return SBN_OK;
}

// ---- profiling: HIP events around every launch on the stream the kernel runs on ----
static int prof_index(sbn_ctx* c, const char* name) {
  for (size_t i = 0; i < c->prof_entries.size(); i++) if (c->prof_entries[i].name == name) return (int)i;
  ProfEntry e; e.name = name; c->prof_entries.push_back(e); return (int)c->prof_entries.size() - 1;
}
static hipEvent_t evt_get(sbn_ctx* c) {
  if (!c->evt_pool.empty()) { hipEvent_t e = c->evt_pool.back(); c->evt_pool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
static void prof_drain(sbn_ctx* c) {
  for (auto& p : c->prof_pending) {
    hipEventSynchronize(p.e1);
    float ms = 0; hipEventElapsedTime(&ms, p.e0, p.e1);
    c->prof_entries[p.idx].ms += ms; c->prof_entries[p.idx].launches += 1;
    c->evt_pool.push_back(p.e0); c->evt_pool.push_back(p.e1);
  }
  c->prof_pending.clear();
}
struct ProfScope {
  sbn_ctx* c; PendingEvt pe; bool on;
  ProfScope(sbn_ctx* c_, const char* name) : c(c_), on(c_->prof) {
    if (on) { pe.idx = prof_index(c, name); pe.e0 = evt_get(c); pe.e1 = evt_get(c); hipEventRecord(pe.e0, c->stream); }
  }
  ~ProfScope() { if (on) { hipEventRecord(pe.e1, c->stream); c->prof_pending.push_back(pe); } }
};
#define LAUNCH(c, name, kern, grid, block, ...) \
  do { ProfScope _ps((c), name); hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, (c)->stream, __VA_ARGS__); } while (0)
#define LAUNCHCHK(c) HIPCHK(c, hipGetLastError())

// ------------------------------------------------------------------------------------------------
static int ilog2_ceil(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }

// Signed radix-2^c digits: W windows cover 254 bits, the top digit (+ carry) must stay <= 2^(c-1).
static MsmShape make_shape(int c) {
  MsmShape s; s.c = c; s.nb = 1 << (c - 1);
  int W = (254 + c - 1) / c;
  int tb = 254 - (W - 1) * c;          // bits in the top window
  if (tb > c - 1) W += 1;
  s.W = W;
  return s;
}
// Window size from a cost model in modular products: `terms`*W mixed adds (10 each) into `sets` bucket sets of 2^(c-1)
// buckets, each bucket costing ~2 full adds (14 each) in the running-sum reduction (x2 for the wave-level part).
// SBN_MSM_C overrides for experiments.
static MsmShape choose_shape(size_t terms, bool shared_bucket_set, int cmax) {
  const char* env = getenv("SBN_MSM_C");
  if (env && atoi(env) >= 7 && atoi(env) <= 22) return make_shape(atoi(env));
  double best = 1e300; int bc = 7;
  // cmax: one sort block keeps all 2^(c-1) counters of a problem in LDS; beyond that every block re-reads its digits once
  // per counter range (measured at 2^26, c = 20: sort 82 ms vs accumulate 74 ms), which costs more than the 13 -> 16 windows.
  for (int c = 7; c <= cmax; c++) {
    MsmShape s = make_shape(c);
    double sets = shared_bucket_set ? 1.0 : (double)s.W;
    double cost = (double)terms * s.W * 10.0 + sets * s.nb * 56.0;
    if (cost < best) { best = cost; bc = c; }
  }
  return make_shape(bc);
}

struct BucketJob {
  int mode; DigitArgs da; MsmShape s;
  size_t P;               // problems (windows or rows)
  size_t threads;         // digit-kernel threads
  const uint32_t* points; // Montgomery affine points the entries index
};

// digits -> counting sort -> segmented bucket accumulation -> per-problem weighted sums in c->wsum (P x XYZZ)
static int run_bucket_job(sbn_ctx* c, const BucketJob& J) {
  const MsmShape& s = J.s;
  const size_t NB = J.P * (size_t)s.nb;
  if (NB > 0xffffffffull) return fail(c, SBN_EINVAL, "bucket space too large");
  const size_t estride = J.da.estride;
  // segment length: twice the mean bucket load (power of two, >= 32)
  size_t mean = estride / (size_t)s.nb + 1;
  uint32_t SEG = 32; while (SEG < 2 * mean && SEG < ACC_SEG_MAX) SEG <<= 1;
  // enough segments to fill the chip when a problem has few, heavily loaded buckets (one row, many columns)
  if (NB < 262144) { const size_t total = J.P * estride; uint32_t cap = 32; while ((size_t)cap * 262144 < total && cap < ACC_SEG_MAX) cap <<= 1; if (SEG > cap) SEG = cap; }
  if (const char* es = getenv("SBN_MSM_SEG")) { int v = atoi(es); if (v >= 8 && v <= (int)ACC_SEG_MAX) SEG = (uint32_t)v; }
  const size_t max_extra = J.P * estride / SEG + 1;
  const size_t max_big = std::min(NB, max_extra);
  int rc;
  if ((rc = ensure(c, c->hist, NB * 4))) return rc;
  if ((rc = ensure(c, c->offs, NB * 4))) return rc;
  if ((rc = ensure(c, c->cursor, NB * 4))) return rc;
  if ((rc = ensure(c, c->sorted, J.P * estride * 4))) return rc;
  if ((rc = ensure(c, c->buckets, NB * 128))) return rc;
  if ((rc = ensure(c, c->acc_ctr, 64))) return rc;
  if ((rc = ensure(c, c->extra_list, max_extra * sizeof(ExtraItem)))) return rc;
  if ((rc = ensure(c, c->extra_out, max_extra * 128))) return rc;
  if ((rc = ensure(c, c->big_list, max_big * sizeof(BigItem)))) return rc;
  // buckets per lane in the reduction: few buckets -> short lane chains and more waves (latency-bound regime); many buckets ->
  // longer chains amortise the wave-level scan/tree (throughput-bound regime).  Aim for ~2048 waves.
  int L = 1; while ((size_t)L * 64 * 2048 < NB && L < 16) L <<= 1;
  if (L < 4) L = 4;
  if (L > s.nb / 64) L = s.nb / 64;
  if (L < 1) L = 1;
  if (const char* el = getenv("SBN_RED_L")) { int v = atoi(el); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0 && v <= s.nb / 64) L = v; }
  int logL = 0; while ((1 << logL) < L) logL++;
  const int chunks = s.nb / (64 * L);                     // per problem, >= 1
  if ((rc = ensure(c, c->red_a, J.P * chunks * 256))) return rc;
  if ((rc = ensure(c, c->red_b, J.P * ((chunks + 63) / 64) * 256))) return rc;
  if ((rc = ensure(c, c->wsum, J.P * 128))) return rc;

  uint32_t* hist = (uint32_t*)c->hist.p; uint32_t* offs = (uint32_t*)c->offs.p; uint32_t* cursor = (uint32_t*)c->cursor.p;
  uint32_t* sorted = (uint32_t*)c->sorted.p; uint32_t* buckets = (uint32_t*)c->buckets.p;
  AccCounters* ctr = (AccCounters*)c->acc_ctr.p;

  HIPCHK(c, hipMemsetAsync(ctr, 0, sizeof(AccCounters), c->stream));
  // digits once, then the LDS counting sort
  SortGeom g; memset(&g, 0, sizeof g);
  g.E = estride; g.estride = estride; g.nb = s.nb; g.mode = J.mode; g.ncol = J.da.n; g.tstride = J.da.tstride;
  g.RS = std::min(s.nb, c->sort_rs_max); g.logRS = 0; while ((1 << g.logRS) < g.RS) g.logRS++;
  g.R = s.nb / g.RS;
  { size_t want = (1024 + J.P * g.R - 1) / (J.P * g.R); size_t maxk = std::max<size_t>(1, estride / 4096); g.K = (int)std::max<size_t>(1, std::min(want, maxk)); }
  g.chunk = (estride + g.K - 1) / g.K;
  if (J.P > 65535 || g.R > 65535) return fail(c, SBN_EINVAL, "sort grid too large (P=%zu R=%d)", J.P, g.R);
  if ((rc = ensure(c, c->digits, J.P * estride * 4))) return rc;
  if ((rc = ensure(c, c->blockhist, J.P * (size_t)g.R * g.K * g.RS * 4))) return rc;
  int32_t* dig = (int32_t*)c->digits.p; uint32_t* bh = (uint32_t*)c->blockhist.p;
  const unsigned gd = (unsigned)((J.threads + 255) / 256);
  if (J.mode == MODE_SINGLE) LAUNCH(c, "k_digits_store", (k_digits_store<MODE_SINGLE>), gd, 256, J.da, s, dig);
  else LAUNCH(c, "k_digits_store", (k_digits_store<MODE_ROWS>), gd, 256, J.da, s, dig);
  if (c->z_consumed && J.mode == MODE_ROWS) HIPCHK(c, hipEventRecord(c->z_consumed, c->stream));   // the scalars are not read again
  const size_t rows_lds = sort_rows_lds_bytes(s.nb);
  if (J.mode == MODE_ROWS && c->sort_rows_ok && rows_lds <= 160 * 1024 && estride <= 8 * (size_t)SORT_SL && !getenv("SBN_NO_FUSED_SORT")) {
    ProfScope _ps(c, "k_sort_rows");
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)J.P), dim3(1024), rows_lds, c->stream, (const int32_t*)dig, g, hist, offs, sorted);
  } else {
    {
      ProfScope _ps(c, "k_hist_lds");
      hipLaunchKernelGGL(k_hist_lds, dim3(g.K, g.R, (unsigned)J.P), dim3(1024), (size_t)g.RS * 4, c->stream, (const int32_t*)dig, g, bh);
    }
    LAUNCH(c, "k_block_prefix", k_block_prefix, (unsigned)((NB + 255) / 256), 256, bh, g, NB, hist);
    LAUNCH(c, "k_scan", k_scan, (unsigned)J.P, 1024, hist, offs, cursor, s.nb);
    {
      ProfScope _ps(c, "k_scatter_lds");
      hipLaunchKernelGGL(k_scatter_lds, dim3(g.K, g.R, (unsigned)J.P), dim3(1024), (size_t)g.RS * 4, c->stream, (const int32_t*)dig, g, (const uint32_t*)bh, (const uint32_t*)offs, sorted);
    }
  }
  // bucket order by decreasing load
  if ((rc = ensure(c, c->size_bins, (ACC_SEG_MAX + 2) * 4))) return rc;
  if ((rc = ensure(c, c->perm, NB * 4))) return rc;
  HIPCHK(c, hipMemsetAsync(c->size_bins.p, 0, (ACC_SEG_MAX + 2) * 4, c->stream));
  LAUNCH(c, "k_size_sort", k_size_hist, (unsigned)((NB + 1023) / 1024), 1024, hist, NB, SEG, (uint32_t*)c->size_bins.p);
  LAUNCH(c, "k_size_sort", k_size_scan, 1, 64, (uint32_t*)c->size_bins.p, SEG);
  LAUNCH(c, "k_size_sort", k_size_scatter, (unsigned)((NB + 1023) / 1024), 1024, hist, NB, SEG, (uint32_t*)c->size_bins.p, (uint32_t*)c->perm.p);
  LAUNCH(c, "k_acc_first", k_acc_first, (unsigned)((NB + 255) / 256), 256, J.points, NB, s.nb, estride, SEG, hist, offs, sorted, (const uint32_t*)c->perm.p, buckets, ctr,
         (ExtraItem*)c->extra_list.p, (BigItem*)c->big_list.p);
  LAUNCH(c, "k_acc_extra", k_acc_extra, 2048, 256, J.points, s.nb, estride, SEG, hist, offs, sorted, ctr, (const ExtraItem*)c->extra_list.p, (uint32_t*)c->extra_out.p);
  LAUNCH(c, "k_acc_merge", k_acc_merge_few, 1024, 256, ctr, (const BigItem*)c->big_list.p, (const uint32_t*)c->extra_out.p, buckets);
  LAUNCH(c, "k_acc_merge", k_acc_merge, 4096, 64, ctr, (const BigItem*)c->big_list.p, (const uint32_t*)c->extra_out.p, buckets);
  LAUNCH(c, "k_reduce_l1", k_reduce_l1, (unsigned)(J.P * chunks), 64, buckets, L, logL, (uint32_t*)c->red_a.p);
  uint32_t* in = (uint32_t*)c->red_a.p; uint32_t* outb = (uint32_t*)c->red_b.p;
  int G = chunks, logM = 6 + logL;
  for (;;) {
    int Gout = (G + 63) / 64;
    int final = (Gout == 1);
    LAUNCH(c, "k_reduce_combine", k_reduce_combine, (unsigned)(J.P * Gout), 64, in, G, Gout, logM, final, final ? (uint32_t*)c->wsum.p : outb);
    if (final) break;
    std::swap(in, outb); G = Gout; logM += 6;
  }
  LAUNCHCHK(c);
  return SBN_OK;
}

// MSM over device-resident canonical scalars and Montgomery affine bases -> canonical affine bytes on the host
static int msm_device(sbn_ctx* c, const uint32_t* d_scal, const uint32_t* d_bases, size_t n, uint8_t out_xy[64], int* out_is_inf) {
  if (n == 0) { memset(out_xy, 0, 64); if (out_is_inf) *out_is_inf = 1; return SBN_OK; }
  if (n > 0x7fffffffull) return fail(c, SBN_EINVAL, "msm: n=%zu exceeds 2^31-1", n);
  BucketJob J; memset(&J, 0, sizeof J);
  J.mode = MODE_SINGLE; { int cm = 1; while ((1 << cm) < c->sort_rs_max) cm++; J.s = choose_shape(n, false, cm + 1); } J.P = (size_t)J.s.W; J.threads = n; J.points = d_bases;
  J.da.scalars = d_scal; J.da.n = n; J.da.estride = n;
  int rc;
  if ((rc = ensure_pin(c, std::max<size_t>(4096, J.P * 128)))) return rc;
  if ((rc = run_bucket_job(c, J))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->pin, c->wsum.p, J.P * 128, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  // sum_w 2^(c w) S_w: the 254-doubling serial chain, on the host
  const sbn_host::Pt* S = (const sbn_host::Pt*)c->pin;
  sbn_host::Pt total = sbn_host::combine_windows(S, J.s.W, J.s.c);
  sbn_host::to_affine_bytes(total, out_xy, out_is_inf);
  return SBN_OK;
}

// window table 2^(c w) * P_j of a generator set, built on first use for a given c and kept with the handle
static int bases_window_table(sbn_ctx* c, const sbn_bases* b, const MsmShape& s, const uint32_t** out) {
  auto it = b->tables.find(s.c);
  if (it != b->tables.end()) { *out = (const uint32_t*)it->second; return SBN_OK; }
  const size_t npts = b->n + (b->has_h ? 1 : 0);
  const size_t tot = npts * (size_t)s.W;
  int rc;
  if ((rc = ensure(c, c->gen_tmp, tot * 128))) return rc;
  void* tab = nullptr;
  hipError_t e = hipMalloc(&tab, tot * 64);
  if (e != hipSuccess) return fail(c, SBN_ENOMEM, "hipMalloc window table (%zu B): %s", tot * 64, hipGetErrorString(e));
  LAUNCH(c, "k_window_table", k_window_table, (unsigned)((npts + 63) / 64), 64, (const uint32_t*)b->d_pts, npts, s.c, s.W, (uint32_t*)c->gen_tmp.p);
  LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((tot + 63) / 64), 64, (const uint32_t*)c->gen_tmp.p, (uint32_t*)tab, (uint32_t*)nullptr, (uint8_t*)nullptr, tot);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  b->tables[s.c] = tab;
  *out = (const uint32_t*)tab;
  return SBN_OK;
}

// Hyrax row commits on device-resident canonical scalars (hyrax.rs:253-267 -> commitments.rs:144-154)
// launches only (no host synchronisation): row commitments as canonical affine bytes + infinity flags in DEVICE buffers
static int commit_rows_launch(sbn_ctx* c, const sbn_bases* b, const uint32_t* dZ, const uint32_t* dBl, size_t L, size_t R, uint32_t* d_xy, uint8_t* d_inf) {
  if (L == 0) return SBN_OK;
  if (b->uniq) {
    // merge the scalars of equal bases, then commit over the unique bases (no blind column: h is merged like any base)
    const size_t U = b->U; int rc;
    if ((rc = ensure(c, c->merged, L * U * 32))) return rc;
    uint32_t* m = (uint32_t*)c->merged.p;
    LAUNCH(c, "k_merge_scalars", k_merge_small, (unsigned)((L * U + 255) / 256), 256, dZ, dBl, L, R, U, (const uint32_t*)b->d_csr_off, (const uint32_t*)b->d_csr_cols, MERGE_BIG, m);
    if (b->nbig) LAUNCH(c, "k_merge_scalars", k_merge_big, (unsigned)(L * b->nbig), 64, dZ, dBl, L, R, U, (const uint32_t*)b->d_csr_off, (const uint32_t*)b->d_csr_cols, (const uint32_t*)b->d_big, b->nbig, m);
    if (c->z_consumed) HIPCHK(c, hipEventRecord(c->z_consumed, c->stream));      // Z (and the blinds) are not read after this point
    return commit_rows_launch(c, b->uniq, m, nullptr, L, U, d_xy, d_inf);
  }
  const size_t ncol = R + (dBl ? 1 : 0);
  if (ncol == 0) { HIPCHK(c, hipMemsetAsync(d_xy, 0, 64 * L, c->stream)); HIPCHK(c, hipMemsetAsync(d_inf, 1, L, c->stream)); return SBN_OK; }
  BucketJob J; memset(&J, 0, sizeof J);
  J.mode = MODE_ROWS; J.s = choose_shape(ncol, true, 16); J.P = L; J.threads = L * ncol;
  const size_t npts = b->n + (b->has_h ? 1 : 0);
  if ((size_t)J.s.W * npts > 0x7fffffffull) return fail(c, SBN_EINVAL, "commit: table index overflow");
  int rc; const uint32_t* tab;
  if ((rc = bases_window_table(c, b, J.s, &tab))) return rc;
  J.points = tab;
  J.da.scalars = dZ; J.da.blinds = dBl; J.da.n = ncol; J.da.R = R; J.da.L = L; J.da.tstride = npts; J.da.estride = ncol * (size_t)J.s.W;
  if ((rc = run_bucket_job(c, J))) return rc;
  LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((L + 63) / 64), 64, (const uint32_t*)c->wsum.p, (uint32_t*)nullptr, d_xy, d_inf, L);
  LAUNCHCHK(c);
  return SBN_OK;
}
// Hyrax row commits on device-resident canonical scalars (hyrax.rs:253-267 -> commitments.rs:144-154)
static int commit_rows_device(sbn_ctx* c, const sbn_bases* b, const uint32_t* dZ, const uint32_t* dBl, size_t L, size_t R, uint8_t* out_xy, uint8_t* out_inf) {
  if (L == 0) return SBN_OK;
  int rc;
  if ((rc = ensure(c, c->out_small, L * 65))) return rc;
  if ((rc = ensure_pin(c, std::max<size_t>(4096, L * 65)))) return rc;
  if ((rc = commit_rows_launch(c, b, dZ, dBl, L, R, (uint32_t*)c->out_small.p, (uint8_t*)c->out_small.p + L * 64))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->pin, c->out_small.p, L * 65, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  memcpy(out_xy, c->pin, L * 64);
  if (out_inf) memcpy(out_inf, (uint8_t*)c->pin + L * 64, L);
  return SBN_OK;
}

static int stage_scalars(sbn_ctx* c, const uint8_t* host_scalars, size_t n, uint32_t flags, const uint32_t** d_out) {
  int rc;
  if ((rc = ensure(c, c->stage_scal, n * 32))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->stage_scal.p, host_scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  if (flags & SBN_SCALARS_MONT) {
    if ((rc = ensure(c, c->scal_canon, n * 32))) return rc;
    LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((n + 255) / 256), 256, (const uint32_t*)c->stage_scal.p, (uint32_t*)c->scal_canon.p, n);
    *d_out = (const uint32_t*)c->scal_canon.p;
  } else *d_out = (const uint32_t*)c->stage_scal.p;
  return SBN_OK;
}
static int canon_scalars_dev(sbn_ctx* c, const void* d_scalars, size_t n, uint32_t flags, const uint32_t** d_out) {
  if (flags & SBN_SCALARS_MONT) {
    int rc; if ((rc = ensure(c, c->scal_canon, n * 32))) return rc;
    LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((n + 255) / 256), 256, (const uint32_t*)d_scalars, (uint32_t*)c->scal_canon.p, n);
    *d_out = (const uint32_t*)c->scal_canon.p;
  } else *d_out = (const uint32_t*)d_scalars;
  return SBN_OK;
}

// Detect equal bases (keys: one byte string per point, equal keys <=> equal points) and, when enough of them repeat, attach
// the unique-point table + CSR column lists used by the commit path.
static int bases_build_dedupe(sbn_ctx* c, sbn_bases* b, const std::vector<std::string>& keys) {
  const size_t tot = keys.size();
  std::unordered_map<std::string, uint32_t> idx;
  std::vector<uint32_t> umap(tot);
  std::vector<uint32_t> first_col;
  for (size_t j = 0; j < tot; j++) {
    auto it = idx.find(keys[j]);
    if (it == idx.end()) { uint32_t u = (uint32_t)first_col.size(); idx.emplace(keys[j], u); first_col.push_back((uint32_t)j); umap[j] = u; }
    else umap[j] = it->second;
  }
  const size_t U = first_col.size();
  if (getenv("SBN_NO_DEDUPE") || U * 10 > tot * 9) return SBN_OK;         // < 10 % repeats: not worth the extra pass
  std::vector<uint32_t> off(U + 1, 0), cols(tot), big;
  for (size_t j = 0; j < tot; j++) off[umap[j] + 1]++;
  for (size_t u = 0; u < U; u++) off[u + 1] += off[u];
  { std::vector<uint32_t> cur(off.begin(), off.end() - 1); for (size_t j = 0; j < tot; j++) cols[cur[umap[j]]++] = (uint32_t)j; }
  for (size_t u = 0; u < U; u++) if (off[u + 1] - off[u] > MERGE_BIG) big.push_back((uint32_t)u);
  sbn_bases* q = new sbn_bases(); q->n = U; q->has_h = false;
  hipError_t e = hipMalloc(&q->d_pts, U * 64);
  if (e != hipSuccess) { delete q; return fail(c, SBN_ENOMEM, "hipMalloc unique bases: %s", hipGetErrorString(e)); }
  for (size_t u = 0; u < U; u++)
    HIPCHK(c, hipMemcpyAsync((uint8_t*)q->d_pts + 64 * u, (const uint8_t*)b->d_pts + 64 * (size_t)first_col[u], 64, hipMemcpyDeviceToDevice, c->stream));
  auto up = [&](void** dst, const std::vector<uint32_t>& v) -> int {
    hipError_t e2 = hipMalloc(dst, std::max<size_t>(4, v.size() * 4)); if (e2 != hipSuccess) return SBN_ENOMEM;
    if (!v.empty() && hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return SBN_EHIP;
    return SBN_OK;
  };
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc;
  if ((rc = up(&b->d_csr_off, off)) || (rc = up(&b->d_csr_cols, cols)) || (rc = up(&b->d_big, big))) { sbn_bases_free(c, q); return fail(c, rc, "dedupe tables"); }
  b->uniq = q; b->U = U; b->nbig = (uint32_t)big.size();
  return SBN_OK;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* sbn_version(void) { return "sbn254-hip 0.1 (gfx950)"; }

int sbn_ctx_create(int device, sbn_ctx** out) {
  if (!out) return SBN_EINVAL;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return SBN_ENODEV;
  if (device < 0 || device >= count) return SBN_ENODEV;
  if (hipSetDevice(device) != hipSuccess) return SBN_ENODEV;
  sbn_ctx* c = new sbn_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SBN_EHIP; }
  c->stream = c->own_stream;
  // 128 KiB of dynamic LDS per sort block (32768 counters); gfx950 has 160 KiB per CU
  if (hipFuncSetAttribute((const void*)k_hist_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4) == hipSuccess) c->sort_rs_max = 32768;
  else (void)hipGetLastError();
  if (hipFuncSetAttribute((const void*)k_sort_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess) c->sort_rows_ok = true;
  else (void)hipGetLastError();
  *out = c;
  return SBN_OK;
}
void sbn_ctx_destroy(sbn_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  prof_drain(c);
  DevBuf* bufs[] = {&c->scal_canon, &c->pts_mont, &c->hist, &c->offs, &c->cursor, &c->sorted, &c->buckets, &c->red_a, &c->red_b, &c->wsum, &c->stage_scal, &c->stage_pts, &c->out_small,
                    &c->sc_args, &c->sc_partial, &c->sc_out, &c->sc_r, &c->sc_tabs, &c->gen_tmp, &c->acc_ctr, &c->extra_list, &c->extra_out, &c->big_list, &c->digits, &c->blockhist, &c->size_bins, &c->perm, &c->merged, &c->zstage[0], &c->zstage[1], &c->out_rows};
  for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
  if (c->pin) hipHostFree(c->pin);
  for (hipEvent_t e : c->evt_pool) hipEventDestroy(e);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  hipStreamDestroy(c->own_stream);
  delete c;
}
const char* sbn_last_error(const sbn_ctx* c) { return c ? c->err.c_str() : "null context"; }
int sbn_ctx_set_stream(sbn_ctx* c, void* s) { if (!c) return SBN_EINVAL; std::lock_guard<std::mutex> g(c->mu); c->stream = s ? (hipStream_t)s : c->own_stream; return SBN_OK; }
int sbn_ctx_sync(sbn_ctx* c) { if (!c) return SBN_EINVAL; std::lock_guard<std::mutex> g(c->mu); HIPCHK(c, hipStreamSynchronize(c->stream)); if (c->prof) prof_drain(c); return SBN_OK; }

int sbn_dev_alloc(sbn_ctx* c, size_t bytes, void** out) { if (!c || !out) return SBN_EINVAL; hipSetDevice(c->device); hipError_t e = hipMalloc(out, bytes ? bytes : 1); if (e != hipSuccess) return fail(c, SBN_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return SBN_OK; }
int sbn_dev_free(sbn_ctx* c, void* p) { if (!c) return SBN_EINVAL; HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(p)); return SBN_OK; }
int sbn_dev_upload(sbn_ctx* c, void* dst, const void* src, size_t bytes) { if (!c || (!dst && bytes) || (!src && bytes)) return SBN_EINVAL; std::lock_guard<std::mutex> g(c->mu); HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return SBN_OK; }
int sbn_dev_download(sbn_ctx* c, void* dst, const void* src, size_t bytes) { if (!c || (!dst && bytes) || (!src && bytes)) return SBN_EINVAL; std::lock_guard<std::mutex> g(c->mu); HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return SBN_OK; }

int sbn_bases_upload(sbn_ctx* c, const uint8_t* G_xy, size_t n, const uint8_t* h_xy, uint32_t flags, sbn_bases** out) {
  if (!c || !out || (!G_xy && n)) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  const size_t tot = n + (h_xy ? 1 : 0);
  sbn_bases* b = new sbn_bases(); b->n = n; b->has_h = h_xy != nullptr;
  hipError_t e = hipMalloc(&b->d_pts, (tot ? tot : 1) * 64);
  if (e != hipSuccess) { delete b; return fail(c, SBN_ENOMEM, "hipMalloc bases: %s", hipGetErrorString(e)); }
  if (n) HIPCHK(c, hipMemcpyAsync(b->d_pts, G_xy, n * 64, hipMemcpyHostToDevice, c->stream));
  if (h_xy) HIPCHK(c, hipMemcpyAsync((uint8_t*)b->d_pts + n * 64, h_xy, 64, hipMemcpyHostToDevice, c->stream));
  if (!(flags & SBN_POINTS_MONT) && tot)
    LAUNCH(c, "k_points_to_mont", k_points_to_mont, (unsigned)((tot + 255) / 256), 256, (const uint32_t*)b->d_pts, (uint32_t*)b->d_pts, tot);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  {
    std::vector<std::string> keys(tot);
    for (size_t j = 0; j < n; j++) keys[j].assign((const char*)G_xy + 64 * j, 64);
    if (h_xy) keys[n].assign((const char*)h_xy, 64);
    int rc = bases_build_dedupe(c, b, keys);
    if (rc) { sbn_bases_free(c, b); return rc; }
  }
  *out = b;
  return SBN_OK;
}
void sbn_bases_free(sbn_ctx* c, sbn_bases* b) { if (!b) return; if (b->uniq) { sbn_bases_free(c, b->uniq); b->uniq = nullptr; }
  if (b->d_csr_off) hipFree(b->d_csr_off); if (b->d_csr_cols) hipFree(b->d_csr_cols); if (b->d_big) hipFree(b->d_big); if (c) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); } if (b->d_pts) hipFree(b->d_pts); for (auto& kv : b->tables) hipFree(kv.second); delete b; }
size_t sbn_bases_len(const sbn_bases* b) { return b ? b->n : 0; }

int sbn_msm_bases_dev(sbn_ctx* c, const sbn_bases* b, const void* d_scalars, size_t n, uint32_t flags, uint8_t out_xy[64], int* out_is_inf) {
  if (!c || !b || !out_xy || (!d_scalars && n)) return SBN_EINVAL;
  if (n > b->n + (b->has_h ? 1 : 0)) return fail(c, SBN_EINVAL, "msm: n=%zu exceeds the table (%zu)", n, b->n + (b->has_h ? 1 : 0));
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  const uint32_t* ds; int rc;
  if ((rc = canon_scalars_dev(c, d_scalars, n, flags, &ds))) return rc;
  return msm_device(c, ds, (const uint32_t*)b->d_pts, n, out_xy, out_is_inf);
}
int sbn_msm_bases(sbn_ctx* c, const sbn_bases* b, const uint8_t* scalars, size_t n, uint32_t flags, uint8_t out_xy[64], int* out_is_inf) {
  if (!c || !b || !out_xy || (!scalars && n)) return SBN_EINVAL;
  if (n > b->n + (b->has_h ? 1 : 0)) return fail(c, SBN_EINVAL, "msm: n=%zu exceeds the table (%zu)", n, b->n + (b->has_h ? 1 : 0));
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  if (n == 0) { memset(out_xy, 0, 64); if (out_is_inf) *out_is_inf = 1; return SBN_OK; }
  const uint32_t* ds; int rc;
  if ((rc = stage_scalars(c, scalars, n, flags, &ds))) return rc;
  return msm_device(c, ds, (const uint32_t*)b->d_pts, n, out_xy, out_is_inf);
}
int sbn_msm(sbn_ctx* c, const uint8_t* scalars, const uint8_t* points, size_t n, uint32_t flags, uint8_t out_xy[64], int* out_is_inf) {
  if (!c || !out_xy || ((!scalars || !points) && n)) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  if (n == 0) { memset(out_xy, 0, 64); if (out_is_inf) *out_is_inf = 1; return SBN_OK; }
  int rc; const uint32_t* ds;
  if ((rc = stage_scalars(c, scalars, n, flags, &ds))) return rc;
  if ((rc = ensure(c, c->stage_pts, n * 64))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->stage_pts.p, points, n * 64, hipMemcpyHostToDevice, c->stream));
  if (!(flags & SBN_POINTS_MONT))
    LAUNCH(c, "k_points_to_mont", k_points_to_mont, (unsigned)((n + 255) / 256), 256, (const uint32_t*)c->stage_pts.p, (uint32_t*)c->stage_pts.p, n);
  return msm_device(c, ds, (const uint32_t*)c->stage_pts.p, n, out_xy, out_is_inf);
}

int sbn_commit_rows_dev(sbn_ctx* c, const sbn_bases* b, const void* Z_dev, const void* blinds_dev, size_t L, size_t R, uint32_t flags, uint8_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || (!out_xy && L) || (!Z_dev && L * R)) return SBN_EINVAL;
  if (R != b->n) return fail(c, SBN_EINVAL, "commit: gens_n.n (%zu) != row length (%zu)  [commitments.rs:146 assert_eq]", b->n, R);
  if (blinds_dev && !b->has_h) return fail(c, SBN_EINVAL, "commit: blinds given but the table has no h");
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  const uint32_t* dZ = (const uint32_t*)Z_dev; const uint32_t* dB = (const uint32_t*)blinds_dev;
  if (flags & SBN_SCALARS_MONT) {
    int rc;
    if ((rc = ensure(c, c->scal_canon, (L * R + L) * 32))) return rc;
    uint32_t* o = (uint32_t*)c->scal_canon.p;
    if (L * R) LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((L * R + 255) / 256), 256, dZ, o, L * R);
    dZ = o;
    if (dB) { LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((L + 255) / 256), 256, dB, o + 8 * L * R, L); dB = o + 8 * L * R; }
  }
  return commit_rows_device(c, b, dZ, dB, L, R, out_xy, out_inf);
}
// Host-pointer variant.  A large matrix is cut into row chunks: while chunk i is being committed, chunk i+1 crosses PCIe
// into the other of two staging buffers (copy stream), so the 1 GiB of a keyless derefs commitment costs about
// max(transfer, compute) instead of their sum.  Rows are independent (hyrax.rs:259-261), so chunking cannot change results.
int sbn_commit_rows(sbn_ctx* c, const sbn_bases* b, const uint8_t* Z, const uint8_t* blinds, size_t L, size_t R, uint32_t flags, uint8_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || (!out_xy && L) || (!Z && L * R)) return SBN_EINVAL;
  if (R != b->n) return fail(c, SBN_EINVAL, "commit: gens_n.n (%zu) != row length (%zu)  [commitments.rs:146 assert_eq]", b->n, R);
  if (blinds && !b->has_h) return fail(c, SBN_EINVAL, "commit: blinds given but the table has no h");
  if (L == 0) return SBN_OK;
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  int rc;
  const size_t row_bytes = R * 32;
  size_t chunk = L;
  size_t chunk_mb = 128;   // measured on the 1 GiB derefs matrix: 64 MB 35.0 ms, 128 MB 28.8, 256 MB 29.4, 512 MB 32.4 (one shot: 40.9)
  if (const char* ec = getenv("SBN_COMMIT_CHUNK_MB")) { int v = atoi(ec); if (v >= 1 && v <= 4096) chunk_mb = (size_t)v; }
  if (L * row_bytes > ((size_t)160 << 20) && row_bytes) { chunk = (chunk_mb << 20) / row_bytes; if (chunk < 1) chunk = 1; }
  const size_t nchunks = (L + chunk - 1) / chunk;
  if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if ((rc = ensure(c, c->out_rows, L * 65))) return rc;
  if ((rc = ensure_pin(c, std::max<size_t>(4096, L * 65)))) return rc;
  for (int k = 0; k < (nchunks > 1 ? 2 : 1); k++) if ((rc = ensure(c, c->zstage[k], chunk * row_bytes + 64))) return rc;
  const uint32_t* dB_all = nullptr;
  if (blinds) {
    if ((rc = ensure(c, c->stage_scal, L * 64))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->stage_scal.p, blinds, L * 32, hipMemcpyHostToDevice, c->stream));
    if (flags & SBN_SCALARS_MONT) {
      uint32_t* o = (uint32_t*)c->stage_scal.p + 8 * L;
      LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((L + 255) / 256), 256, (const uint32_t*)c->stage_scal.p, o, L);
      dB_all = o;
    } else dB_all = (const uint32_t*)c->stage_scal.p;
  }
  hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
  for (int k = 0; k < 2; k++) { copied[k] = evt_get(c); consumed[k] = evt_get(c); }
  uint32_t* d_xy = (uint32_t*)c->out_rows.p; uint8_t* d_inf = (uint8_t*)c->out_rows.p + L * 64;
  rc = SBN_OK;
  for (size_t i = 0; i < nchunks && rc == SBN_OK; i++) {
    const int k = (int)(i & 1);
    const size_t r0 = i * chunk, rows = std::min(chunk, L - r0);
    // the staging buffer is free again once the chunk that used it two iterations ago has been read by its first kernels
    if (i >= 2) { hipError_t e = hipStreamWaitEvent(c->copy_stream, consumed[k], 0); if (e != hipSuccess) { rc = fail(c, SBN_EHIP, "hipStreamWaitEvent: %s", hipGetErrorString(e)); break; } }
    hipError_t e = hipMemcpyAsync(c->zstage[k].p, Z + r0 * row_bytes, rows * row_bytes, hipMemcpyHostToDevice, c->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(copied[k], c->copy_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, copied[k], 0);
    if (e != hipSuccess) { rc = fail(c, SBN_EHIP, "chunk upload: %s", hipGetErrorString(e)); break; }
    const uint32_t* dZ = (const uint32_t*)c->zstage[k].p;
    if (flags & SBN_SCALARS_MONT) {
      if ((rc = ensure(c, c->scal_canon, chunk * row_bytes))) break;
      LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((rows * R + 255) / 256), 256, dZ, (uint32_t*)c->scal_canon.p, rows * R);
      hipEventRecord(consumed[k], c->stream);
      dZ = (const uint32_t*)c->scal_canon.p;
      c->z_consumed = nullptr;
    } else c->z_consumed = consumed[k];
    rc = commit_rows_launch(c, b, dZ, dB_all ? dB_all + 8 * r0 : nullptr, rows, R, d_xy + 16 * r0, d_inf + r0);
    c->z_consumed = nullptr;
  }
  if (rc == SBN_OK) {
    hipError_t e = hipMemcpyAsync(c->pin, c->out_rows.p, L * 65, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) rc = fail(c, SBN_EHIP, "result download: %s", hipGetErrorString(e));
  }
  hipStreamSynchronize(c->copy_stream);
  hipError_t es = hipStreamSynchronize(c->stream);
  if (rc == SBN_OK && es != hipSuccess) rc = fail(c, SBN_EHIP, "commit: %s", hipGetErrorString(es));
  for (int k = 0; k < 2; k++) { c->evt_pool.push_back(copied[k]); c->evt_pool.push_back(consumed[k]); }
  if (c->prof) prof_drain(c);
  if (rc) return rc;
  memcpy(out_xy, c->pin, L * 64);
  if (out_inf) memcpy(out_inf, (uint8_t*)c->pin + L * 64, L);
  return SBN_OK;
}

int sbn_g1_compress(const uint8_t* xy, size_t n, uint8_t* out32) {
  if ((!xy || !out32) && n) return SBN_EINVAL;
  for (size_t i = 0; i < n; i++) {
    const uint8_t* p = xy + 64 * i; uint8_t* o = out32 + 32 * i;
    bool inf = true; for (int k = 0; k < 64; k++) if (p[k]) { inf = false; break; }
    if (inf) { memset(o, 0, 32); o[31] = 0x40; continue; }
    memcpy(o, p, 32);
    // y > p - y  <=>  2y > p
    uint64_t y[4], t[4]; memcpy(y, p + 32, 32);
    uint64_t cy = 0; for (int k = 0; k < 4; k++) { t[k] = (y[k] << 1) | cy; cy = y[k] >> 63; }
    bool gt = cy != 0;
    if (!gt) { gt = false; for (int k = 3; k >= 0; k--) { if (t[k] > sbn_host::QP[k]) { gt = true; break; } if (t[k] < sbn_host::QP[k]) break; } }
    if (gt) o[31] |= 0x80;
  }
  return SBN_OK;
}
int sbn_g1_sum(const uint8_t* xy, size_t n, uint8_t out_xy[64], int* out_is_inf) {
  if (!out_xy || (!xy && n)) return SBN_EINVAL;
  sbn_host::Pt acc = sbn_host::inf();
  for (size_t i = 0; i < n; i++) {
    const uint8_t* p = xy + 64 * i;
    bool inf = true; for (int k = 0; k < 64; k++) if (p[k]) { inf = false; break; }
    if (inf) continue;
    sbn_host::Fq x, y; memcpy(x.v, p, 32); memcpy(y.v, p + 32, 32);
    if (sbn_host::geq_p(x.v) || sbn_host::geq_p(y.v)) return SBN_EINVAL;   // not canonical
    sbn_host::Pt q; q.X = sbn_host::to_mont(x); q.Y = sbn_host::to_mont(y); q.ZZ = sbn_host::one(); q.ZZZ = sbn_host::one();
    acc = sbn_host::padd(acc, q);
  }
  sbn_host::to_affine_bytes(acc, out_xy, out_is_inf);
  return SBN_OK;
}
void sbn_factored_lens(size_t ell, size_t* left, size_t* right) { if (left) *left = ell / 2; if (right) *right = ell - ell / 2; }

// ---- generators: MultiCommitGens::new (commitments.rs:31-62) ----
static const uint64_t FR_MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static bool fr_canonical(const uint8_t b[32]) {
  uint64_t v[4]; memcpy(v, b, 32);
  for (int i = 3; i >= 0; i--) { if (v[i] < FR_MOD[i]) return true; if (v[i] > FR_MOD[i]) return false; }
  return false;
}
// GroupElement::from_uniform_bytes (group.rs:110-131): the scalar s with point = s*G
static void uniform_bytes_scalar(const uint8_t ub[64], uint8_t s[32]) {
  sbn_host::sha3_256(ub, 64, s);
  if (fr_canonical(s)) return;                 // Scalar::from_bytes accepts only < r (scalar.rs:87-95)
  uint8_t tmp[72]; memcpy(tmp, "fallback", 8); memcpy(tmp + 8, ub, 64);
  sbn_host::sha3_256(tmp, 72, s);
  if (fr_canonical(s)) return;
  memset(s, 0, 32); s[0] = 1;                  // unwrap_or(Scalar::one())
}
int sbn_gens_new(sbn_ctx* c, size_t n, const uint8_t* label, size_t label_len, uint8_t* out_xy, sbn_bases** out) {
  if (!c || !out || (!label && label_len)) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  const size_t tot = n + 1;
  // SHAKE256(label || compressed generator); the generator (1,2) compresses to 01 00..00 (y = 2 is the smaller root)
  uint8_t gc[32] = {1};
  sbn_host::Keccak xof(136, 0x1f);
  xof.absorb(label, label_len); xof.absorb(gc, 32);
  std::vector<uint8_t> dl(tot * 32);
  for (size_t i = 0; i < tot; i++) { uint8_t ub[64]; xof.squeeze(ub, 64); uniform_bytes_scalar(ub, &dl[32 * i]); }
  int rc;
  if ((rc = ensure(c, c->gen_tmp, tot * (32 + 128)))) return rc;
  uint8_t* d_s = (uint8_t*)c->gen_tmp.p; uint8_t* d_x = d_s + tot * 32;
  HIPCHK(c, hipMemcpyAsync(d_s, dl.data(), tot * 32, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_mul_generator", k_mul_generator, (unsigned)((tot + 63) / 64), 64, (const uint32_t*)d_s, tot, (uint32_t*)d_x);
  sbn_bases* b = new sbn_bases(); b->n = n; b->has_h = true;
  hipError_t e = hipMalloc(&b->d_pts, tot * 64);
  if (e != hipSuccess) { delete b; return fail(c, SBN_ENOMEM, "hipMalloc gens: %s", hipGetErrorString(e)); }
  uint32_t* d_xy = nullptr;
  if (out_xy) { if ((rc = ensure(c, c->out_small, tot * 64))) { hipFree(b->d_pts); delete b; return rc; } d_xy = (uint32_t*)c->out_small.p; }
  LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((tot + 63) / 64), 64, (const uint32_t*)d_x, (uint32_t*)b->d_pts, d_xy, (uint8_t*)nullptr, tot);
  LAUNCHCHK(c);
  if (out_xy) HIPCHK(c, hipMemcpyAsync(out_xy, d_xy, tot * 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  {
    std::vector<std::string> keys(tot);     // equal discrete logs <=> equal points
    for (size_t j = 0; j < tot; j++) keys[j].assign((const char*)&dl[32 * j], 32);
    if ((rc = bases_build_dedupe(c, b, keys))) { sbn_bases_free(c, b); return rc; }
  }
  *out = b;
  return SBN_OK;
}

int sbn_msm_jacobian(sbn_ctx* c, const uint8_t* scalars, const uint8_t* points_xyz, size_t n, uint32_t flags, uint8_t out_xy[64], int* out_is_inf) {
  if (!c || !out_xy || ((!scalars || !points_xyz) && n)) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  if (n == 0) { memset(out_xy, 0, 64); if (out_is_inf) *out_is_inf = 1; return SBN_OK; }
  int rc; const uint32_t* ds;
  if ((rc = stage_scalars(c, scalars, n, flags, &ds))) return rc;
  if ((rc = ensure(c, c->gen_tmp, n * 96))) return rc;
  if ((rc = ensure(c, c->stage_pts, n * 64))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->gen_tmp.p, points_xyz, n * 96, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_jacobian_to_affine", k_jacobian_to_affine, (unsigned)((n + 63) / 64), 64, (const uint32_t*)c->gen_tmp.p, (flags & SBN_POINTS_MONT) ? 1 : 0, n, (uint32_t*)c->stage_pts.p);
  return msm_device(c, ds, (const uint32_t*)c->stage_pts.p, n, out_xy, out_is_inf);
}
// device-side copy of a point range into a new handle (keys for the duplicate detection are read back once)
static int bases_from_device(sbn_ctx* c, const void* d_G, size_t n, const void* d_h, sbn_bases** out) {
  const size_t tot = n + (d_h ? 1 : 0);
  sbn_bases* b = new sbn_bases(); b->n = n; b->has_h = d_h != nullptr;
  hipError_t e = hipMalloc(&b->d_pts, (tot ? tot : 1) * 64);
  if (e != hipSuccess) { delete b; return fail(c, SBN_ENOMEM, "hipMalloc bases: %s", hipGetErrorString(e)); }
  if (n) HIPCHK(c, hipMemcpyAsync(b->d_pts, d_G, n * 64, hipMemcpyDeviceToDevice, c->stream));
  if (d_h) HIPCHK(c, hipMemcpyAsync((uint8_t*)b->d_pts + n * 64, d_h, 64, hipMemcpyDeviceToDevice, c->stream));
  std::vector<std::string> keys(tot);
  std::vector<uint8_t> host(tot * 64);
  if (tot) HIPCHK(c, hipMemcpyAsync(host.data(), b->d_pts, tot * 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t j = 0; j < tot; j++) keys[j].assign((const char*)&host[64 * j], 64);
  int rc = bases_build_dedupe(c, b, keys);
  if (rc) { sbn_bases_free(c, b); return rc; }
  *out = b;
  return SBN_OK;
}
int sbn_bases_split_at(sbn_ctx* c, const sbn_bases* b, size_t mid, sbn_bases** left, sbn_bases** right) {
  if (!c || !b || !left || !right || mid > b->n) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  const uint8_t* p = (const uint8_t*)b->d_pts; const void* h = b->has_h ? p + 64 * b->n : nullptr;
  int rc;
  if ((rc = bases_from_device(c, p, mid, h, left))) return rc;
  if ((rc = bases_from_device(c, p + 64 * mid, b->n - mid, h, right))) { sbn_bases_free(c, *left); *left = nullptr; return rc; }
  return SBN_OK;
}
int sbn_bases_scale(sbn_ctx* c, const sbn_bases* b, const uint8_t s[32], sbn_bases** out) {
  if (!c || !b || !s || !out) return SBN_EINVAL;
  if (!fr_canonical(s)) return fail(c, SBN_EINVAL, "scale: scalar not canonical");
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  int rc; const size_t n = b->n;
  if ((rc = ensure(c, c->gen_tmp, 64 + n * 128 + n * 64))) return rc;
  uint8_t* d_s = (uint8_t*)c->gen_tmp.p; uint8_t* d_x = d_s + 64; uint8_t* d_a = d_x + n * 128;
  HIPCHK(c, hipMemcpyAsync(d_s, s, 32, hipMemcpyHostToDevice, c->stream));
  if (n) {
    LAUNCH(c, "k_scale_points", k_scale_points, (unsigned)((n + 63) / 64), 64, (const uint32_t*)b->d_pts, n, (const uint32_t*)d_s, (uint32_t*)d_x);
    LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((n + 63) / 64), 64, (const uint32_t*)d_x, (uint32_t*)d_a, (uint32_t*)nullptr, (uint8_t*)nullptr, n);
  }
  LAUNCHCHK(c);
  return bases_from_device(c, d_a, n, b->has_h ? (const uint8_t*)b->d_pts + 64 * n : nullptr, out);
}
int sbn_bases_synthetic(sbn_ctx* c, size_t n, uint64_t first, const uint8_t s0[32], const uint8_t d[32], sbn_bases** out) {
  if (!c || !out || !s0 || !d || n == 0) return SBN_EINVAL;
  if (!fr_canonical(s0) || !fr_canonical(d)) return fail(c, SBN_EINVAL, "synthetic bases: s0/d not canonical");
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  int rc;
  if ((rc = ensure(c, c->gen_tmp, 64 + 256 + n * 128))) return rc;
  uint8_t* d_s = (uint8_t*)c->gen_tmp.p; uint8_t* d_p0d = d_s + 64; uint8_t* d_x = d_p0d + 256;
  HIPCHK(c, hipMemcpyAsync(d_s, s0, 32, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_s + 32, d, 32, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_mul_generator", k_mul_generator, 1, 64, (const uint32_t*)d_s, (size_t)2, (uint32_t*)d_p0d);
  LAUNCH(c, "k_arith_points", k_arith_points, (unsigned)((n + 63) / 64), 64, (const uint32_t*)d_p0d, (unsigned long long)first, n, (uint32_t*)d_x);
  sbn_bases* b = new sbn_bases(); b->n = n; b->has_h = false;
  hipError_t e = hipMalloc(&b->d_pts, n * 64);
  if (e != hipSuccess) { delete b; return fail(c, SBN_ENOMEM, "hipMalloc synthetic bases: %s", hipGetErrorString(e)); }
  LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((n + 63) / 64), 64, (const uint32_t*)d_x, (uint32_t*)b->d_pts, (uint32_t*)nullptr, (uint8_t*)nullptr, n);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out = b;
  return SBN_OK;
}
int sbn_bases_download(sbn_ctx* c, const sbn_bases* b, size_t first, size_t count, uint8_t* out_xy) {
  if (!c || !b || (!out_xy && count)) return SBN_EINVAL;
  const size_t tot = b->n + (b->has_h ? 1 : 0);
  if (first > tot || count > tot - first) return fail(c, SBN_EINVAL, "bases_download: range outside the table");
  if (count == 0) return SBN_OK;
  std::lock_guard<std::mutex> g(c->mu);
  hipSetDevice(c->device);
  int rc; if ((rc = ensure(c, c->stage_pts, count * 64))) return rc;
  LAUNCH(c, "k_points_from_mont", k_points_from_mont, (unsigned)((count + 255) / 256), 256, (const uint32_t*)b->d_pts + 16 * first, (uint32_t*)c->stage_pts.p, count);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(out_xy, c->stage_pts.p, count * 64, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SBN_OK;
}

// ---- tables + sumcheck rounds ----
static unsigned stream_grid(size_t work_items) {
  size_t blocks = (work_items + 255) / 256;
  if (blocks > 2048) blocks = 2048;     // 256 CUs x 8 blocks, grid-stride beyond
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
static int table_make(sbn_ctx* c, const void* src, bool src_is_host, size_t len, uint32_t flags, sbn_table** out) {
  if (len == 0 || (len & (len - 1))) return fail(c, SBN_EINVAL, "table length %zu is not a power of two", len);
  sbn_table* t = new sbn_table(); t->len = len; t->cap = len;
  hipError_t e = hipMalloc(&t->d, len * 32);
  if (e != hipSuccess) { delete t; return fail(c, SBN_ENOMEM, "hipMalloc table: %s", hipGetErrorString(e)); }
  HIPCHK(c, hipMemcpyAsync(t->d, src, len * 32, src_is_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
  if (!(flags & SBN_SCALARS_MONT)) LAUNCH(c, "k_fr_to_mont", k_fr_to_mont, stream_grid(len), 256, (const uint32_t*)t->d, (uint32_t*)t->d, len);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out = t;
  return SBN_OK;
}
int sbn_table_upload(sbn_ctx* c, const uint8_t* Z, size_t len, uint32_t flags, sbn_table** out) {
  if (!c || !Z || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  return table_make(c, Z, true, len, flags, out);
}
int sbn_table_from_dev(sbn_ctx* c, const void* Z_dev, size_t len, uint32_t flags, sbn_table** out) {
  if (!c || !Z_dev || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  return table_make(c, Z_dev, false, len, flags, out);
}
void sbn_table_free(sbn_ctx* c, sbn_table* t) { if (!t) return; if (c) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); } if (t->d && t->owned) hipFree(t->d); if (t->d2 && t->owned2) hipFree(t->d2); delete t; }
size_t sbn_table_len(const sbn_table* t) { return t ? t->len : 0; }
int sbn_table_download(sbn_ctx* c, const sbn_table* t, uint8_t* out) {
  if (!c || !t || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  int rc; if ((rc = ensure(c, c->stage_scal, t->len * 32))) return rc;
  LAUNCH(c, "k_fr_from_mont", k_fr_from_mont, stream_grid(t->len), 256, (const uint32_t*)t->d, (uint32_t*)c->stage_scal.p, t->len);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(out, c->stage_scal.p, t->len * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SBN_OK;
}
int sbn_table_read0(sbn_ctx* c, const sbn_table* t, uint8_t out[32]) {
  if (!c || !t || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  int rc; if ((rc = ensure(c, c->sc_out, 4096))) return rc;
  LAUNCH(c, "k_fr_from_mont", k_fr_from_mont, 1, 256, (const uint32_t*)t->d, (uint32_t*)c->sc_out.p, (size_t)1);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(out, c->sc_out.p, 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SBN_OK;
}
static int upload_r_mont(sbn_ctx* c, const uint8_t r[32]) {
  int rc; if ((rc = ensure(c, c->sc_r, 64))) return rc;
  if ((rc = ensure_pin(c, 4096))) return rc;
  if (!fr_canonical(r)) return fail(c, SBN_EINVAL, "challenge scalar is not canonical (>= r)");
  memcpy(c->pin, r, 32);
  HIPCHK(c, hipMemcpyAsync(c->sc_r.p, c->pin, 32, hipMemcpyHostToDevice, c->stream));
  LAUNCH(c, "k_fr_to_mont", k_fr_to_mont, 1, 256, (const uint32_t*)c->sc_r.p, (uint32_t*)c->sc_r.p, (size_t)1);
  return SBN_OK;
}
int sbn_bind_top_many(sbn_ctx* c, sbn_table* const* ts, size_t count, const uint8_t r[32]) {
  if (!c || !ts || !r || count == 0) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  for (size_t i = 0; i < count; i++) { if (!ts[i]) return SBN_EINVAL; if (ts[i]->len != ts[0]->len) return fail(c, SBN_EINVAL, "bind: tables differ in length"); }
  if (ts[0]->len < 2) return fail(c, SBN_EINVAL, "bind: table has no variable left");
  int rc; if ((rc = upload_r_mont(c, r))) return rc;
  if ((rc = ensure(c, c->sc_tabs, count * sizeof(void*)))) return rc;
  // pinned staging lives after the 32-byte r slot
  if ((rc = ensure_pin(c, 4096 + count * sizeof(void*)))) return rc;
  void** hp = (void**)((uint8_t*)c->pin + 64);
  for (size_t i = 0; i < count; i++) hp[i] = ts[i]->d;
  HIPCHK(c, hipMemcpyAsync(c->sc_tabs.p, hp, count * sizeof(void*), hipMemcpyHostToDevice, c->stream));
  const size_t half = ts[0]->len / 2;
  unsigned gx = stream_grid(half); if (count > 1 && gx > 1024) gx = 1024;
  LAUNCH(c, "k_bind_top", k_bind_top, dim3(gx, (unsigned)count), 256, (uint32_t* const*)c->sc_tabs.p, half, (const uint32_t*)c->sc_r.p);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));      // the pinned staging is reused by the next call
  if (c->prof) prof_drain(c);
  for (size_t i = 0; i < count; i++) ts[i]->len = half;
  return SBN_OK;
}
int sbn_bind_top(sbn_ctx* c, sbn_table* t, const uint8_t r[32]) { sbn_table* one[1] = {t}; return sbn_bind_top_many(c, one, 1, r); }

}  // extern "C" (templates need C++ linkage)
template <int KIND>
static int sc_eval_common(sbn_ctx* c, const sbn_table* const* const* cols, int ncols, size_t count, uint8_t* out) {
  // cols[j][i] = table j of instance i
  const size_t len = cols[0][0]->len;
  for (int j = 0; j < ncols; j++) for (size_t i = 0; i < count; i++) {
    if (!cols[j][i]) return SBN_EINVAL;
    if (cols[j][i]->len != len) return fail(c, SBN_EINVAL, "sumcheck eval: tables differ in length");
  }
  if (len < 2) return fail(c, SBN_EINVAL, "sumcheck eval: no variable left");
  const size_t half = len / 2;
  int rc;
  if ((rc = ensure(c, c->sc_args, count * sizeof(ScArgs)))) return rc;
  if ((rc = ensure_pin(c, 4096 + count * sizeof(ScArgs) + count * 96))) return rc;
  ScArgs* ha = (ScArgs*)((uint8_t*)c->pin + 64);
  for (size_t i = 0; i < count; i++) for (int j = 0; j < 4; j++) ha[i].t[j] = j < ncols ? (const uint32_t*)cols[j][i]->d : nullptr;
  HIPCHK(c, hipMemcpyAsync(c->sc_args.p, ha, count * sizeof(ScArgs), hipMemcpyHostToDevice, c->stream));
  unsigned gx = stream_grid(half); if (gx > 1024) gx = 1024;
  if ((rc = ensure(c, c->sc_partial, (size_t)count * gx * 96))) return rc;
  if ((rc = ensure(c, c->sc_out, std::max<size_t>(4096, count * 96)))) return rc;
  const char* nm = KIND == KIND_CUBIC ? "k_sc_eval_cubic" : KIND == KIND_R1CS ? "k_sc_eval_r1cs" : "k_sc_eval_quad";
  LAUNCH(c, nm, k_sc_eval<KIND>, dim3(gx, (unsigned)count), 256, (const ScArgs*)c->sc_args.p, half, (uint32_t*)c->sc_partial.p);
  LAUNCH(c, "k_sc_finish", k_sc_finish, (unsigned)count, 64, (const uint32_t*)c->sc_partial.p, (int)gx, (uint32_t*)c->sc_out.p);
  LAUNCHCHK(c);
  uint8_t* hres = (uint8_t*)c->pin + 64 + count * sizeof(ScArgs);
  HIPCHK(c, hipMemcpyAsync(hres, c->sc_out.p, count * 96, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  if (KIND == KIND_QUAD) { for (size_t i = 0; i < count; i++) memcpy(out + 64 * i, hres + 96 * i, 64); }
  else memcpy(out, hres, count * 96);
  return SBN_OK;
}
extern "C" {
int sbn_sc_eval_cubic_batched(sbn_ctx* c, const sbn_table* const* A, const sbn_table* const* B, const sbn_table* const* Cc, size_t count, uint8_t* out) {
  if (!c || !A || !B || !Cc || !out || count == 0) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  const sbn_table* const* cols[3] = {A, B, Cc};
  return sc_eval_common<KIND_CUBIC>(c, cols, 3, count, out);
}
int sbn_sc_eval_cubic(sbn_ctx* c, const sbn_table* A, const sbn_table* B, const sbn_table* Cc, uint8_t out[96]) {
  if (!A || !B || !Cc) return SBN_EINVAL;
  return sbn_sc_eval_cubic_batched(c, &A, &B, &Cc, 1, out);
}
int sbn_sc_eval_r1cs(sbn_ctx* c, const sbn_table* T, const sbn_table* A, const sbn_table* B, const sbn_table* Cc, uint8_t out[96]) {
  if (!c || !T || !A || !B || !Cc || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  const sbn_table* const* cols[4] = {&T, &A, &B, &Cc};
  return sc_eval_common<KIND_R1CS>(c, cols, 4, 1, out);
}
int sbn_sc_eval_quad(sbn_ctx* c, const sbn_table* Z, const sbn_table* ABC, uint8_t out[64]) {
  if (!c || !Z || !ABC || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  const sbn_table* const* cols[2] = {&Z, &ABC};
  return sc_eval_common<KIND_QUAD>(c, cols, 2, 1, out);
}
}  // extern "C"
template <int KIND>
static int sc_bind_eval_common(sbn_ctx* c, sbn_table* const* const* cols, int ncols, size_t count, const uint8_t r[32], uint8_t* out) {
  const size_t len = cols[0][0]->len;
  for (int j = 0; j < ncols; j++) for (size_t i = 0; i < count; i++) {
    if (!cols[j][i]) return SBN_EINVAL;
    if (cols[j][i]->len != len) return fail(c, SBN_EINVAL, "sumcheck bind+eval: tables differ in length");
  }
  if (len < 4) return fail(c, SBN_EINVAL, "sumcheck bind+eval needs len >= 4 (use sbn_bind_top for the last round)");
  const size_t q = len / 4;
  int rc;
  if ((rc = upload_r_mont(c, r))) return rc;
  if ((rc = ensure(c, c->sc_args, count * sizeof(ScFusedArgs)))) return rc;
  if ((rc = ensure_pin(c, 4096 + count * sizeof(ScFusedArgs) + count * 96))) return rc;
  // every distinct table gets exactly one writer; its second buffer receives the bound half
  std::vector<sbn_table*> distinct;
  ScFusedArgs* ha = (ScFusedArgs*)((uint8_t*)c->pin + 64);
  for (size_t i = 0; i < count; i++) for (int j = 0; j < 4; j++) {
    ha[i].src[j] = nullptr; ha[i].dst[j] = nullptr;
    if (j >= ncols) continue;
    sbn_table* t = cols[j][i];
    ha[i].src[j] = (const uint32_t*)t->d;
    if (std::find(distinct.begin(), distinct.end(), t) == distinct.end()) {
      if (t->cap2 < len / 2) {
        if (t->d2) { HIPCHK(c, hipStreamSynchronize(c->stream)); if (t->owned2) HIPCHK(c, hipFree(t->d2)); t->d2 = nullptr; t->cap2 = 0; }
        t->owned2 = true;
        hipError_t e = hipMalloc(&t->d2, (len / 2) * 32);
        if (e != hipSuccess) return fail(c, SBN_ENOMEM, "hipMalloc second table buffer: %s", hipGetErrorString(e));
        t->cap2 = len / 2;
      }
      ha[i].dst[j] = (uint32_t*)t->d2;
      distinct.push_back(t);
    }
  }
  HIPCHK(c, hipMemcpyAsync(c->sc_args.p, ha, count * sizeof(ScFusedArgs), hipMemcpyHostToDevice, c->stream));
  unsigned gx = stream_grid(q); if (gx > 1024) gx = 1024;
  if ((rc = ensure(c, c->sc_partial, (size_t)count * gx * 96))) return rc;
  if ((rc = ensure(c, c->sc_out, std::max<size_t>(4096, count * 96)))) return rc;
  const char* nm = KIND == KIND_CUBIC ? "k_sc_bind_eval_cubic" : KIND == KIND_R1CS ? "k_sc_bind_eval_r1cs" : "k_sc_bind_eval_quad";
  LAUNCH(c, nm, k_sc_bind_eval<KIND>, dim3(gx, (unsigned)count), 256, (const ScFusedArgs*)c->sc_args.p, q, (const uint32_t*)c->sc_r.p, (uint32_t*)c->sc_partial.p);
  LAUNCH(c, "k_sc_finish", k_sc_finish, (unsigned)count, 64, (const uint32_t*)c->sc_partial.p, (int)gx, (uint32_t*)c->sc_out.p);
  LAUNCHCHK(c);
  uint8_t* hres = (uint8_t*)c->pin + 64 + count * sizeof(ScFusedArgs);
  HIPCHK(c, hipMemcpyAsync(hres, c->sc_out.p, count * 96, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  for (sbn_table* t : distinct) { std::swap(t->d, t->d2); std::swap(t->cap, t->cap2); std::swap(t->owned, t->owned2); t->len = len / 2; }
  if (KIND == KIND_QUAD) { for (size_t i = 0; i < count; i++) memcpy(out + 64 * i, hres + 96 * i, 64); }
  else memcpy(out, hres, count * 96);
  return SBN_OK;
}
extern "C" {
int sbn_sc_bind_eval_cubic_batched(sbn_ctx* c, sbn_table* const* A, sbn_table* const* B, sbn_table* const* Cc, size_t count, const uint8_t r[32], uint8_t* out) {
  if (!c || !A || !B || !Cc || !r || !out || count == 0) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  sbn_table* const* cols[3] = {A, B, Cc};
  return sc_bind_eval_common<KIND_CUBIC>(c, cols, 3, count, r, out);
}
int sbn_sc_bind_eval_r1cs(sbn_ctx* c, sbn_table* T, sbn_table* A, sbn_table* B, sbn_table* Cc, const uint8_t r[32], uint8_t out[96]) {
  if (!c || !T || !A || !B || !Cc || !r || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  sbn_table* const* cols[4] = {&T, &A, &B, &Cc};
  return sc_bind_eval_common<KIND_R1CS>(c, cols, 4, 1, r, out);
}
int sbn_sc_bind_eval_quad(sbn_ctx* c, sbn_table* Z, sbn_table* ABC, const uint8_t r[32], uint8_t out[64]) {
  if (!c || !Z || !ABC || !r || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  sbn_table* const* cols[2] = {&Z, &ABC};
  return sc_bind_eval_common<KIND_QUAD>(c, cols, 2, 1, r, out);
}
int sbn_hash_layer(sbn_ctx* c, const void* addr_dev, const sbn_table* val, const void* ts_dev, uint32_t ts_add, const uint8_t r_hash[32], const uint8_t r_multiset[32], sbn_table** out) {
  if (!c || !val || !r_hash || !r_multiset || !out) return SBN_EINVAL;
  if (!fr_canonical(r_hash) || !fr_canonical(r_multiset)) return fail(c, SBN_EINVAL, "hash layer: challenges not canonical");
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  const size_t n = val->len; int rc;
  if ((rc = ensure(c, c->sc_r, 64 + 96))) return rc;
  if ((rc = ensure_pin(c, 4096))) return rc;
  memcpy(c->pin, r_hash, 32); memcpy((uint8_t*)c->pin + 32, r_multiset, 32);
  HIPCHK(c, hipMemcpyAsync(c->sc_r.p, c->pin, 64, hipMemcpyHostToDevice, c->stream));
  uint32_t* consts = (uint32_t*)c->sc_r.p + 16;
  LAUNCH(c, "k_hash_consts", k_hash_consts, 1, 64, (const uint32_t*)c->sc_r.p, consts);
  sbn_table* t = new sbn_table(); t->len = n; t->cap = n;
  hipError_t e = hipMalloc(&t->d, n * 32);
  if (e != hipSuccess) { delete t; return fail(c, SBN_ENOMEM, "hipMalloc hash layer: %s", hipGetErrorString(e)); }
  LAUNCH(c, "k_hash_layer", k_hash_layer, stream_grid(n), 256, (const uint32_t*)addr_dev, (const uint32_t*)val->d, (const uint32_t*)ts_dev, ts_add, (const uint32_t*)consts, n, (uint32_t*)t->d);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  *out = t;
  return SBN_OK;
}
int sbn_product_layer(sbn_ctx* c, const sbn_table* in, sbn_table** out) {
  if (!c || !in || !out) return SBN_EINVAL;
  if (in->len < 2) return fail(c, SBN_EINVAL, "product layer: nothing left to multiply");
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  const size_t half = in->len / 2;
  sbn_table* t = new sbn_table(); t->len = half; t->cap = half;
  hipError_t e = hipMalloc(&t->d, half * 32);
  if (e != hipSuccess) { delete t; return fail(c, SBN_ENOMEM, "hipMalloc product layer: %s", hipGetErrorString(e)); }
  LAUNCH(c, "k_product_layer", k_product_layer, stream_grid(half), 256, (const uint32_t*)in->d, half, (uint32_t*)t->d);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  *out = t;
  return SBN_OK;
}
int sbn_table_halves(sbn_ctx* c, const sbn_table* t, sbn_table** left, sbn_table** right) {
  if (!c || !t || !left || !right) return SBN_EINVAL;
  if (t->len < 2) return fail(c, SBN_EINVAL, "halves: table has one entry");
  const size_t half = t->len / 2;
  sbn_table* l = new sbn_table(); sbn_table* r = new sbn_table();
  l->d = t->d; l->len = l->cap = half; l->owned = false;
  r->d = (uint8_t*)t->d + half * 32; r->len = r->cap = half; r->owned = false;
  *left = l; *right = r;
  return SBN_OK;
}
static int table_dot_locked(sbn_ctx* c, const uint32_t* a, const uint32_t* b, size_t n, uint8_t out[32]) {
  int rc;
  unsigned gx = stream_grid(n); if (gx > 1024) gx = 1024;
  if ((rc = ensure(c, c->sc_partial, (size_t)gx * 96))) return rc;
  if ((rc = ensure(c, c->sc_out, 4096))) return rc;
  if ((rc = ensure_pin(c, 4096))) return rc;
  LAUNCH(c, "k_dot", k_dot, gx, 256, a, b, n, (uint32_t*)c->sc_partial.p);
  LAUNCH(c, "k_sc_finish", k_sc_finish, 1, 64, (const uint32_t*)c->sc_partial.p, (int)gx, (uint32_t*)c->sc_out.p);
  LAUNCHCHK(c);
  HIPCHK(c, hipMemcpyAsync(c->pin, c->sc_out.p, 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  memcpy(out, c->pin, 32);
  return SBN_OK;
}
int sbn_table_dot(sbn_ctx* c, const sbn_table* a, const sbn_table* b, uint8_t out[32]) {
  if (!c || !a || !b || !out) return SBN_EINVAL;
  if (a->len != b->len) return fail(c, SBN_EINVAL, "dot: lengths differ (hyrax.rs:410 assert_eq)");
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  return table_dot_locked(c, (const uint32_t*)a->d, (const uint32_t*)b->d, a->len, out);
}
int sbn_table_evaluate(sbn_ctx* c, const sbn_table* Z, const uint8_t* r, size_t ell, uint8_t out[32]) {
  if (!c || !Z || (!r && ell) || !out) return SBN_EINVAL;
  if (((size_t)1 << ell) != Z->len) return fail(c, SBN_EINVAL, "evaluate: r.len() != num_vars (hyrax.rs:218 assert_eq)");
  sbn_table* chi = nullptr;
  int rc = sbn_eq_evals(c, r, ell, &chi);
  if (rc) return rc;
  rc = sbn_table_dot(c, Z, chi, out);
  sbn_table_free(c, chi);
  return rc;
}
int sbn_table_bound(sbn_ctx* c, const sbn_table* Z, const sbn_table* Lv, sbn_table** out) {
  if (!c || !Z || !Lv || !out) return SBN_EINVAL;
  const size_t L_size = Lv->len;
  if (L_size == 0 || Z->len % L_size) return fail(c, SBN_EINVAL, "bound: table length is not a multiple of L.len()");
  const size_t R_size = Z->len / L_size;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  // row slices so that ~2048 blocks are in flight
  const size_t col_tiles = (R_size + 63) / 64;
  size_t nslices = (2048 + col_tiles - 1) / col_tiles; if (nslices > L_size) nslices = L_size; if (nslices < 1) nslices = 1;
  const size_t rows_per_slice = (L_size + nslices - 1) / nslices; nslices = (L_size + rows_per_slice - 1) / rows_per_slice;
  if (col_tiles > 0x7fffffff || nslices > 65535) return fail(c, SBN_EINVAL, "bound: grid too large");
  int rc;
  if ((rc = ensure(c, c->sc_partial, nslices * R_size * 32))) return rc;
  sbn_table* t = new sbn_table(); t->len = R_size; t->cap = R_size;
  hipError_t e = hipMalloc(&t->d, R_size * 32);
  if (e != hipSuccess) { delete t; return fail(c, SBN_ENOMEM, "hipMalloc bound table: %s", hipGetErrorString(e)); }
  LAUNCH(c, "k_bound_partial", k_bound_partial, dim3((unsigned)col_tiles, (unsigned)nslices), 256, (const uint32_t*)Z->d, (const uint32_t*)Lv->d, L_size, R_size, rows_per_slice, (uint32_t*)c->sc_partial.p);
  LAUNCH(c, "k_bound_fold", k_bound_fold, (unsigned)((R_size + 255) / 256), 256, (const uint32_t*)c->sc_partial.p, nslices, R_size, (uint32_t*)t->d);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  *out = t;
  return SBN_OK;
}
int sbn_gather_merge(sbn_ctx* c, const sbn_table* const* mem, const void* const* addr_dev, size_t count, size_t n, sbn_table** out) {
  if (!c || !mem || !addr_dev || !out || count == 0 || n == 0) return SBN_EINVAL;
  for (size_t k = 0; k < count; k++) if (!mem[k] || !addr_dev[k]) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  size_t padded = 1; while (padded < count * n) padded <<= 1;              // Z.resize(len.next_power_of_two()) (hyrax.rs:245)
  int rc;
  if ((rc = ensure(c, c->sc_args, count * sizeof(GatherArgs) + 64))) return rc;
  if ((rc = ensure_pin(c, 4096 + count * sizeof(GatherArgs)))) return rc;
  GatherArgs* ha = (GatherArgs*)((uint8_t*)c->pin + 64);
  for (size_t k = 0; k < count; k++) { ha[k].mem = (const uint32_t*)mem[k]->d; ha[k].addr = (const uint32_t*)addr_dev[k]; ha[k].mem_len = mem[k]->len; }
  uint8_t* d_args = (uint8_t*)c->sc_args.p; uint32_t* d_oob = (uint32_t*)(d_args + count * sizeof(GatherArgs));
  HIPCHK(c, hipMemcpyAsync(d_args, ha, count * sizeof(GatherArgs), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(d_oob, 0, 4, c->stream));
  sbn_table* t = new sbn_table(); t->len = padded; t->cap = padded;
  hipError_t e = hipMalloc(&t->d, padded * 32);
  if (e != hipSuccess) { delete t; return fail(c, SBN_ENOMEM, "hipMalloc gather table: %s", hipGetErrorString(e)); }
  LAUNCH(c, "k_gather_merge", k_gather_merge, stream_grid(padded), 256, (const GatherArgs*)d_args, count, n, padded, (uint32_t*)t->d, d_oob);
  LAUNCHCHK(c);
  uint32_t oob = 0;
  HIPCHK(c, hipMemcpyAsync(c->pin, d_oob, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  memcpy(&oob, c->pin, 4);
  if (oob) { hipFree(t->d); delete t; return fail(c, SBN_EINVAL, "gather: %u addresses are outside their memory table (sparse_mlpoly_full.rs:228 assert)", oob); }
  *out = t;
  return SBN_OK;
}
int sbn_commit_table(sbn_ctx* c, const sbn_bases* b, const sbn_table* t, const uint8_t* blinds, size_t L, size_t R, uint8_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || !t || (!out_xy && L)) return SBN_EINVAL;
  if (L * R != t->len) return fail(c, SBN_EINVAL, "commit_table: L*R (%zu) != table length (%zu)  [hyrax.rs:257 assert_eq]", L * R, t->len);
  void* dB = nullptr; int rc;
  if (blinds) {
    if ((rc = sbn_dev_alloc(c, L * 32, &dB))) return rc;
    if ((rc = sbn_dev_upload(c, dB, blinds, L * 32))) { sbn_dev_free(c, dB); return rc; }
    // blinds arrive canonical while the table is Montgomery: bring the blinds to Montgomery form so one flag covers both
    std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
    LAUNCH(c, "k_fr_to_mont", k_fr_to_mont, stream_grid(L), 256, (const uint32_t*)dB, (uint32_t*)dB, L);
  }
  rc = sbn_commit_rows_dev(c, b, t->d, dB, L, R, SBN_SCALARS_MONT, out_xy, out_inf);
  if (dB) sbn_dev_free(c, dB);
  return rc;
}
int sbn_eq_evals(sbn_ctx* c, const uint8_t* r, size_t ell, sbn_table** out) {
  if (!c || (!r && ell) || !out || ell > 40) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu); hipSetDevice(c->device);
  for (size_t j = 0; j < ell; j++) if (!fr_canonical(r + 32 * j)) return fail(c, SBN_EINVAL, "eq_evals: r[%zu] is not canonical", j);
  const size_t N = (size_t)1 << ell;
  int rc;
  if ((rc = ensure(c, c->stage_scal, std::max<size_t>(N * 32, 64)))) return rc;     // ping-pong partner
  if ((rc = ensure(c, c->sc_r, std::max<size_t>(64, ell * 32)))) return rc;
  sbn_table* t = new sbn_table(); t->len = N; t->cap = N;
  hipError_t e = hipMalloc(&t->d, N * 32);
  if (e != hipSuccess) { delete t; return fail(c, SBN_ENOMEM, "hipMalloc eq table: %s", hipGetErrorString(e)); }
  if (ell) {
    HIPCHK(c, hipMemcpyAsync(c->sc_r.p, r, ell * 32, hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, "k_fr_to_mont", k_fr_to_mont, 1, 256, (const uint32_t*)c->sc_r.p, (uint32_t*)c->sc_r.p, ell);
  }
  // ping-pong so that the last level lands in t->d
  uint32_t* bufA = (uint32_t*)t->d; uint32_t* bufB = (uint32_t*)c->stage_scal.p;
  uint32_t* cur = (ell % 2 == 0) ? bufA : bufB;
  LAUNCH(c, "k_fr_set_one", k_fr_set_one, 1, 64, cur);
  size_t size = 1;
  for (size_t j = 0; j < ell; j++) {
    uint32_t* nxt = (cur == bufA) ? bufB : bufA;
    LAUNCH(c, "k_eq_level", k_eq_level, stream_grid(size), 256, (const uint32_t*)cur, nxt, size, (const uint32_t*)c->sc_r.p + 8 * j);
    cur = nxt; size *= 2;
  }
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out = t;
  return SBN_OK;
}

int sbn_prof_enable(sbn_ctx* c, int on) { if (!c) return SBN_EINVAL; std::lock_guard<std::mutex> g(c->mu); c->prof = on != 0; return SBN_OK; }
int sbn_prof_reset(sbn_ctx* c) { if (!c) return SBN_EINVAL; std::lock_guard<std::mutex> g(c->mu); hipStreamSynchronize(c->stream); prof_drain(c); c->prof_entries.clear(); return SBN_OK; }
int sbn_prof_count(sbn_ctx* c) { if (!c) return 0; std::lock_guard<std::mutex> g(c->mu); hipStreamSynchronize(c->stream); prof_drain(c); return (int)c->prof_entries.size(); }
int sbn_prof_get(sbn_ctx* c, int i, const char** name, double* total_ms, uint64_t* launches) {
  if (!c || i < 0 || (size_t)i >= c->prof_entries.size()) return SBN_EINVAL;
  if (name) *name = c->prof_entries[i].name.c_str();
  if (total_ms) *total_ms = c->prof_entries[i].ms;
  if (launches) *launches = c->prof_entries[i].launches;
  return SBN_OK;
}

}  // extern "C"
