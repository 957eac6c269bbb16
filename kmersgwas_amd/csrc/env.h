// env.h — every environment switch of libkgwas and its tools, in one place.
//
// OPTIONS (kgwas::opt_*): the product's run-time switches. Each one is documented in the table below, is read where it acts, and
// is walked by the test matrix (tests/, tools/fuzz_parity.py) - none selects code the default tests do not reach:
//
//   KGWAS_FULL_REPLAY=1        every column is replayed push by push (no select mode)
//   KGWAS_COARSE_MX=0|1        filter family: 0 int8 (score_coarse.hip), 1 block-scaled FP4 x FP6/FP4 (score_mx.hip); unset: by plan
//   KGWAS_COARSE_SLICES=1|2    force the one- / two-slice operand set (unset: chosen per chunk / by plan)
//   KGWAS_MX_S1=6              block-scaled filter: FP6 second slice instead of FP4
//   KGWAS_MXS=0..3             operand-streaming form (score_mxs.hip): never / where the resident form degenerates (default) /
//                              wherever it needs > 1 LDS group / wherever the form exists
//   KGWAS_MXS_FORM=1|2         its block shapes with one column group of up to 13 tiles
//   KGWAS_NARROW=0             1-4 columns through the wide filter instead of the narrow one
//   KGWAS_HOST_THREADS=n       replay threads of a session (overrides kgwas_scan_params.host_threads)
//   KGWAS_FINISH_THREADS=n     kgwas_scan_finish makes the select-mode columns' lists on n threads if that is more than the pool
//   KGWAS_PIN_THREADS=0|1|2    replay threads: unpinned / one CPU each / one core (SMT pair) each (default: 1 where the CPU quota covers the CPU set, else 0)
//   KGWAS_SPLIT_LAGGING=0      a column group that falls behind is not cut into single columns
//   KGWAS_FLOAT_LEAD=n         chunks a group may lag before it floats to the workers that are ahead (default 2, 0: never)
//   KGWAS_HISTORY_RING=n       record_history = 2: evictions kept per heap (default 16 sqrt(2 N))
//   KGWAS_RING_BYTES=n         size of the pinned record ring
//   KGWAS_LOG_BY_REF=0         select-mode logs copy their records instead of referring to the record ring
//   KGWAS_RECORD_COPY=memcpy|kernel   how a chunk's records reach the host ring
//   KGWAS_INGEST_PIECE_ROWS / KGWAS_INGEST_PINNED / KGWAS_INGEST_DEVICE   geometry of the streamed feed's piece rings
//   KGWAS_PIN_PLAIN=1          pinned buffers from hipHostMalloc instead of registered huge-page mappings
//   KGWAS_TRACE=1              timeline of a scan on stderr
//   KGWAS_CLI_FULL_TEARDOWN=1  the command-line tools destroy their sessions instead of _exit
//   KGWAS_AUTO_PARALLEL=1      associate_kmers: replay threads = the CPUs the process may use, whatever --parallel says
//   KGWAS_DEVICE=n             associate_snps: device ordinal
//   KGWAS_DEBUG_SLOW_WORKER=w:pct:min_us   (test hook) slows one replay worker down
//   KGWAS_DEBUG_RESIDUALS=1    (test hook) sessions keep their filters' quantisation residuals (kgwas_scan_debug_residuals)
//
// EXPERIMENTS (kgwas::exp_*): tuning and ablation knobs of tools/ (chunk-size policies, block sizes, prefetch distances, older
// forms of a step kept for A/B runs). They exist only in a build with -DKGWAS_EXPERIMENTS (`make EXPERIMENTS=1`); in the
// shipped library every exp_* call is its default - a compile-time constant, no getenv, and the branch it guarded folds away.
#pragma once
#include <stdlib.h>

namespace kgwas {

inline const char* opt_str(const char* name) { return getenv(name); }
inline bool opt_set(const char* name) { return opt_str(name) != nullptr; }
inline long long opt_int(const char* name, long long dflt) {
    const char* e = opt_str(name);
    return e && *e ? atoll(e) : dflt;
}

#ifdef KGWAS_EXPERIMENTS
inline const char* exp_str(const char* name) { return opt_str(name); }
inline bool exp_set(const char* name) { return opt_set(name); }
inline long long exp_int(const char* name, long long dflt) { return opt_int(name, dflt); }
inline double exp_num(const char* name, double dflt) {
    const char* e = opt_str(name);
    return e && *e ? atof(e) : dflt;
}
#else
constexpr const char* exp_str(const char*) { return nullptr; }
constexpr bool exp_set(const char*) { return false; }
constexpr long long exp_int(const char*, long long dflt) { return dflt; }
constexpr double exp_num(const char*, double dflt) { return dflt; }
#endif

}  // namespace kgwas
