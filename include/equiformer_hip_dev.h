/*
 * equiformer_hip_dev.h -- development entry points of libequiformer_hip.so.
 *
 * NOT part of the drop-in boundary (that is equiformer_hip.h): these switches exist for A/B measurements and
 * phase timing of the fused SeparableFCTP and GEMM kernels (tools/sfc_exp.py, tools/bench_sfc.py, tools/sfc_race.py,
 * tools/gemm_exp.py).  They change process-global state, are not thread safe and have no reference counterpart.  Nothing
 * under equiformer_amd/ calls them.
 *
 * The switches that act INSIDE kernels (eqf_sfc_debug_exp bits 1, 2, 4, 8; eqf_sfc_debug_buffer; eqf_gemm_debug_exp bits 1,
 * 2) are compiled in only with -DEQF_DEV_SWITCHES=1 (EQF_EXTRA_FLAGS="-DEQF_DEV_SWITCHES=1" python -m equiformer_amd.build);
 * in the product build they are accepted and do nothing, so the hot loops carry no test of them.  The in-kernel clock
 * samples of csrc/sfcx.hip (eqf_sfcx_dev_set_trace, tools/sfcx_trace.py) exist only in a -DEQF_XTRACE=1 build.
 */
#ifndef EQUIFORMER_HIP_DEV_H
#define EQUIFORMER_HIP_DEV_H

#include "equiformer_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* workgroup ordering of the three sfc kernels: -1 defaults ({fwd 0, bwd_data 1, bwd_weight 1}), 0 x-fastest,
 * 1 XCD-aware, 2 y-fastest */
int eqf_sfc_debug_order(int mode);
/* bit mask that switches phases of the sfc kernels off (1 no MFMA loop, 2 no generation / register epilogue, 4 no
 * re-staging, 8 no weight loads, 16 paired workgroups, 32 force one workgroup per CU, 64 the non-default forward matrix
 * step: split-precision bf16 x 6 -- see the note at g_sfc_x6_default in csrc/sfc.hip) */
int eqf_sfc_debug_exp(int mask);
/* 1 if the split-precision forward step is the default (it is not) */
int eqf_sfc_debug_x6_default(void);
/* 8 x u64 device counters the sfc kernels add per-phase cycle counts to; NULL disables */
int eqf_sfc_debug_buffer(void* device_u64x8);
/* argument tables of the split-precision SeparableFCTP launches (eqf_sfcx_*) as text: kind 0 forward, 1 data gradient,
 * 2 weight gradient (one-wave kernel), 3 weight gradient (multi-wave kernel of csrc/sfcw.hip: workgroup types).  Host-only (no GPU needed): tests/test_sfcx_plan.py replays the kernels' lane-level algorithm in
 * numpy on these tables.  Returns the number of characters written or a negative error. */
int eqf_sfcx_dev_plan(int kind, const eqf_dtp_paths* paths, const eqf_irreps* out1_irreps, int n2, int E, int mode,
                      char* buf, int buflen);
/* switches of the split-precision kernels (process-global, for tests and A/B timing; 0 restores the default unless noted):
 *   key 1   the data gradient runs only the items of input degree (value - 1) / 2, to time one class of work items
 *   key 2   forward: 1 = always the one-wave kernel, 2 = the multi-wave kernel (csrc/sfcy.hip) wherever its planner accepts
 *   key 3   one-wave weight gradient: only the items of class 1 + 10 (2 l1 + 1) + (2 l3 + 1)
 *   key 4   weight gradient: 1 = always the one-wave kernel, 2 = the multi-wave kernel (csrc/sfcw.hip) wherever accepted
 *   key 5 / 6 / 7   multi-wave weight gradient: rounds of resident workgroups the chunk length is sized for / launch order
 *           (0 heaviest first in batches of 8 chunks per XCD, 1 chunk-major, 2 type-major heaviest first) / only the workgroups
 *           of one type (-1: all)
 *   key 9   data gradient on small graphs: 1 = never split the paths of an item over two waves
 *   key 10  ... wave slots the split pairs of a launch may take (default 1 536)
 *   key 11  forward on small graphs: at most this many waves per item (1, 2, 4; default 4)
 * A/B measurements of kernel variants use variant builds of the library (equiformer_amd/build.py --variant NAME -DEQF_...=...) */
int eqf_sfcx_dev_set(int key, int value);
/* csrc/gemmx.hip: key 0 = 0 selects the one-wave-per-tile kernels for node-row problems (default 1: LDS-tiled kernels for all) */
int eqf_gemmx_dev_set(int key, int value);
/* gemm kernels: 1 no stores, 2 no MFMA */
int eqf_gemm_debug_exp(int mask);

#ifdef __cplusplus
}
#endif
#endif /* EQUIFORMER_HIP_DEV_H */
