// ctx.hpp — context, workspace and per-kernel profiling of libsbn254_hip.so (included by sbn254.hip only).
#pragma once
// ------------------------------------------------------------------------------------------------
static const uint32_t SBN_SCALARS_INTERNAL = 0x10000u;   // private flag: the scalars are one of this library's device tables (Montgomery R = 2^261, lazy representatives)
struct ProfEntry { std::string name; double ms = 0; uint64_t launches = 0; };
struct PendingEvt { int idx; hipEvent_t e0, e1; };

struct DevBuf {
  void* p = nullptr; size_t cap = 0;
};

// Experiment overrides of the sumcheck paths (SBN_SC_*: tools/README.md), read ONCE when the context is created — never on a round's path
// (441 rounds per prove; getenv is not safe against a concurrent setenv).  A test or sweep that wants another setting creates a context.
struct ScKnobs {
  bool no_tiny = false, no_mixed = false, no_mixed_eval = false, no_comb = false, no_comb_kernel = false, no_prebind = false, no_stream_mbox = false, no_fuse_c = false;
  size_t comb_grid = 0;              // 0: automatic
  size_t comb_blocks = 512, seq_blocks = 512, comb_eval_blocks_mixed = 768, eval_blocks_mixed = 768, comb_eval_blocks = 1024, eval_blocks = 2048;
  size_t comb_min_q = (size_t)1 << 14, comb_lanes = 262144, single_max = (size_t)1 << 14, grid = 0, block_rounds = 4;
  std::string debug_blocks;          // SBN_SC_DEBUG_BLOCKS=<file>
};
static ScKnobs sc_knobs_read() {
  ScKnobs k;
  auto flag = [](const char* n) { return getenv(n) != nullptr; };
  auto num = [](const char* n, long long lo, size_t* out) { const char* e = getenv(n); if (!e) return false; const long long x = atoll(e); if (x < lo) return false; *out = (size_t)x; return true; };
  k.no_tiny = flag("SBN_SC_NO_TINY"); k.no_mixed = flag("SBN_SC_NO_MIXED"); k.no_mixed_eval = flag("SBN_SC_NO_MIXED_EVAL"); k.no_comb = flag("SBN_SC_NO_COMB");
  k.no_comb_kernel = flag("SBN_SC_NO_COMB_KERNEL"); k.no_prebind = flag("SBN_SC_NO_PREBIND"); k.no_stream_mbox = flag("SBN_SC_NO_STREAM_MBOX"); k.no_fuse_c = flag("SBN_SC_NO_FUSE_C");
  num("SBN_SC_COMB_GRID", 1, &k.comb_grid); num("SBN_SC_COMB_BLOCKS", 1, &k.comb_blocks); num("SBN_SC_SEQ_BLOCKS", 1, &k.seq_blocks);
  if (num("SBN_SC_COMB_EVAL_BLOCKS", 1, &k.comb_eval_blocks)) k.comb_eval_blocks_mixed = k.comb_eval_blocks;
  if (num("SBN_SC_EVAL_BLOCKS", 1, &k.eval_blocks)) k.eval_blocks_mixed = k.eval_blocks;
  num("SBN_SC_COMB_MIN_Q", 1, &k.comb_min_q); num("SBN_SC_COMB_LANES", 1, &k.comb_lanes); num("SBN_SC_SINGLE_MAX", 2, &k.single_max); num("SBN_SC_GRID", 1, &k.grid);
  { size_t v = 0; if (num("SBN_SC_BLOCK_ROUNDS", 1, &v) && v <= 16) k.block_rounds = v; }
  if (const char* e = getenv("SBN_SC_DEBUG_BLOCKS")) k.debug_blocks = e;
  return k;
}

struct sbn_ctx {
  int device = 0;
  ScKnobs sck;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::mutex mu;
  std::string err;
  // workspace (grown on demand, never shrunk; no allocation in steady state)
  DevBuf scal_canon, hist, offs, sorted, buckets, red_a, red_b, wsum, stage_scal, stage_pts, out_small;
  DevBuf sc_args, sc_partial, sc_out, sc_r, sc_tabs, sc_tickets, gen_tmp, acc_ctr, extra_list, extra_out, big_list, digits, blockhist, perm, merged;
  hipStream_t copy_stream = nullptr;          // H2D of the next row chunk while the current one is being committed
  hipEvent_t z_consumed = nullptr;            // set while a chunked commit is running: recorded when a chunk's scalars have been read
  DevBuf zstage[2], out_rows, comb_partial;
  DevBuf s2_cnt, s2_part, s2_idx, s2_lo;     // two-level sort of a large single MSM (sort2_kernels.cuh)
  bool sort2_ok = false;      // dynamic LDS of its level-1 scatter granted
  size_t sort2_min = (size_t)1 << 20;   // terms from which a single MSM takes the two-level sort (SBN_SORT2_MIN; 0 = never)
  int sc_waves = 2;           // streaming sumcheck rounds: 2 = software-pipelined loads, 2 waves per SIMD (default); 3 / 4 = the plain kernel at that occupancy (SBN_SC_WAVES)
  bool sort_rows_ok = false;  // 160 KiB dynamic LDS granted to k_sort_rows
  int sort_rs_max = 16384;   // LDS counters per sort block (raised to 32768 when 128 KiB of dynamic LDS is granted)
  void* pin = nullptr; size_t pin_cap = 0;   // pinned host staging for small D2H results
  uint32_t* d_bad = nullptr;                 // device word: scalars >= r met by the kernels that read the caller's input (input_check_*)
  uint32_t* h_bad = nullptr;                 // its pinned host copy
  uint32_t* mbox = nullptr; uint32_t mbox_seq = 0;   // coherent pinned mailbox of the single-launch sumcheck rounds (results + per-instance flags)
  std::vector<std::pair<void*, size_t>> pool; size_t pool_bytes = 0;   // cached table buffers (see pool_get)
  uint64_t last_job[4] = {0, 0, 0, 0};      // window bits, windows, (digit, point) slots, buckets of the most recent bucket job
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
  size_t U = 0; uint32_t nbig = 0; uint32_t hcol = 0;   // hcol: the unique base h maps to
  void* d_csr_off = nullptr; void* d_csr_cols = nullptr; void* d_big = nullptr;
  // fixed-base direct-lookup table (comb_kernels.cuh): W x npts x 2^(c-1) affine points, built by sbn_bases_precompute
  void* d_comb = nullptr; int comb_c = 0; size_t comb_bytes = 0;
  // bullet reduction (abi_bullet.inc): derived sets G ‖ Q (+ h), one per distinct Q, built on first use and owned by this handle
  mutable std::vector<std::pair<std::string, sbn_bases*>> bullet_ext;
};
static const uint32_t MERGE_BIG = 64;
extern "C" void sbn_bases_free(sbn_ctx* c, sbn_bases* b);

struct sbn_table {
  void* d = nullptr; size_t len = 0; size_t cap = 0; bool owned = true;      // cap / cap2: capacity in 32-byte elements
  void* d2 = nullptr; size_t cap2 = 0; bool owned2 = true;     // second buffer for the fused (out-of-place) bind
};

// Table buffers come from a small per-context cache: hipMalloc / hipFree of a 1 GiB buffer costs tens of milliseconds (it
// synchronises the device and maps / unmaps pages), more than the kernels that fill it, and a prover allocates the same
// shapes every round.  pool_get returns a cached buffer of at least `bytes` (best fit) or allocates; pool_put keeps up to
// POOL_MAX_ENTRIES buffers / POOL_MAX_BYTES and releases the rest.
static const size_t POOL_MAX_ENTRIES = 1024;   // a keyless-sized prove holds ~300 product-circuit layers + their second buffers
static const size_t POOL_MAX_BYTES = (size_t)24 << 30;
static hipError_t pool_get(sbn_ctx* c, size_t bytes, void** out, size_t* got_bytes) {
  if (bytes == 0) bytes = 32;
  int best = -1;
  for (size_t i = 0; i < c->pool.size(); i++)
    if (c->pool[i].second >= bytes && c->pool[i].second <= 2 * bytes && (best < 0 || c->pool[i].second < c->pool[(size_t)best].second)) best = (int)i;
  if (best >= 0) { *out = c->pool[(size_t)best].first; *got_bytes = c->pool[(size_t)best].second; c->pool_bytes -= *got_bytes; c->pool.erase(c->pool.begin() + best); return hipSuccess; }
  hipError_t e = hipMalloc(out, bytes);
  if (e != hipSuccess && !c->pool.empty()) {          // out of memory: drop the cache and retry once
    (void)hipGetLastError();
    for (auto& b : c->pool) hipFree(b.first);
    c->pool.clear(); c->pool_bytes = 0;
    e = hipMalloc(out, bytes);
  }
  *got_bytes = bytes;
  return e;
}
static void pool_put(sbn_ctx* c, void* p, size_t bytes) {
  if (!p) return;
  if (!c || c->pool.size() >= POOL_MAX_ENTRIES || c->pool_bytes + bytes > POOL_MAX_BYTES) { hipFree(p); return; }
  c->pool.emplace_back(p, bytes); c->pool_bytes += bytes;
}

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
// Non-canonical input scalars (>= r): Scalar::from_bytes rejects them (scalar.rs:87-95) and the digit recoding assumes the top
// window cannot carry out, so the kernels that read caller-provided scalars count them in c->d_bad.  An entry point clears the
// counter before its first launch, copies it back with its result and turns a non-zero count into SBN_EINVAL.
static int input_check_begin(sbn_ctx* c) { HIPCHK(c, hipMemsetAsync(c->d_bad, 0, 4, c->stream)); return SBN_OK; }
static int input_check_fetch(sbn_ctx* c) { HIPCHK(c, hipMemcpyAsync(c->h_bad, c->d_bad, 4, hipMemcpyDeviceToHost, c->stream)); return SBN_OK; }   // enqueue before the final synchronisation
static int input_check_end(sbn_ctx* c) {
  if (*c->h_bad) return fail(c, SBN_EINVAL, "%u input scalars are not canonical (>= r; Scalar::from_bytes rejects them, scalar.rs:87-95)", *c->h_bad);
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

// ---- the host mailbox (coherent pinned memory the result kernels write straight into) and its waits ----
// ticket counters of the single-launch rounds: zero at rest (the last block resets its counter)
static int sc_tickets(sbn_ctx* c) {
  if (c->mbox && c->sc_tickets.cap) return SBN_OK;
  if (!c->mbox) {                                  // the mailbox first: a failed allocation must not leave the tickets looking initialised
    uint32_t* mb = nullptr;
    HIPCHK(c, hipHostMalloc((void**)&mb, SC_MBOX_WORDS * 4, hipHostMallocMapped | hipHostMallocCoherent));
    memset(mb, 0, SC_MBOX_WORDS * 4);
    c->mbox = mb;
  }
  if (!c->sc_tickets.cap) {
    int rc; if ((rc = ensure(c, c->sc_tickets, 4096 * 4))) return rc;
    HIPCHK(c, hipMemsetAsync(c->sc_tickets.p, 0, 4096 * 4, c->stream));
  }
  return SBN_OK;
}
// Wait for the `count` flags of launch `seq`: a short spin on the mailbox (the kernel is microseconds long when this path
// matters), then the ordinary stream synchronisation.  With event profiling on, always synchronise (the events must be complete).
static int sc_mbox_wait(sbn_ctx* c, size_t count, uint32_t seq) {
  volatile uint32_t* fl = c->mbox + SC_MBOX_FLAGS;
  if (!c->prof) {
    for (int spin = 0; spin < 200000; spin++) {
      size_t done = 0; while (done < count && fl[done] == seq) done++;
      if (done == count) { std::atomic_thread_fence(std::memory_order_acquire); return SBN_OK; }
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  for (size_t i = 0; i < count; i++) if (fl[i] != seq) return fail(c, SBN_EHIP, "sumcheck round: result flag %zu missing after synchronisation", i);
  return SBN_OK;
}
// one flag word of the mailbox (the stateful sumcheck's final claims): spin briefly, then synchronise
static int sc_flag_wait(sbn_ctx* c, volatile uint32_t* flag, uint32_t seq) {
  if (!c->prof) {
    for (int spin = 0; spin < 200000; spin++) if (*flag == seq) { std::atomic_thread_fence(std::memory_order_acquire); return SBN_OK; }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  if (*flag != seq) return fail(c, SBN_EHIP, "result flag missing in the host mailbox after synchronisation");
  return SBN_OK;
}
