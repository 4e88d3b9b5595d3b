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
#include "sort2_kernels.cuh"
#include "comb_kernels.cuh"
#include "sumcheck_kernels.cuh"
#include "sumcheck_comb_kernels.cuh"
#include "host_keccak.hpp"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace sbn;

#include "ctx.hpp"
#include "msm_host.hpp"

#include "abi_msm.inc"
#include "abi_tables.inc"
#include "abi_sumcheck.inc"
#include "abi_bullet.inc"
#include "abi_group.inc"

extern "C" {

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
int sbn_prof_last_job(sbn_ctx* c, uint64_t out[4]) {
  if (!c || !out) return SBN_EINVAL;
  std::lock_guard<std::mutex> g(c->mu);
  for (int i = 0; i < 4; i++) out[i] = c->last_job[i];
  return SBN_OK;
}

}  // extern "C"
