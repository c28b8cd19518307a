/*
 * mxdet_debug.h -- tuning and test hooks of libmxdet_hip.so. NOT part of the drop-in boundary (include/mxdet.h):
 * nothing a reference-side binding calls is declared here. Unlike the boundary's entries, the hooks set state that
 * later calls read (per calling thread where noted, otherwise library-wide): sweeps and tests only.
 */
#ifndef MXDET_DEBUG_H_
#define MXDET_DEBUG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* force the conv tile configuration on this thread (0 = built-in heuristic, else a case of conv.hip launch()) */
int mxdet_debug_force_conv_cfg(int32_t cfg);
/* force the split-K factor of mxdet_conv2d_wgrad on the calling thread (0 = heuristic) */
int mxdet_debug_force_wgrad_ksplit(int32_t ksplit);
/* 1 = always use the direct-gather preprocess kernel (the wide-frame path); library-wide */
int mxdet_debug_preprocess_direct(int32_t on);

/* Plan-time thresholds of the tile / split heuristics (library-wide; value < 0 restores the built-in default).
 * The library itself never reads the environment: tools that sweep these call the hook. */
#define MXDET_TUNE_T64 0          /* conv: 64-row tiles from this many tiles on (default 400) */
#define MXDET_TUNE_T128 1         /* conv: 256x256 (+ tail) tiles from this many 128x128 tiles on (default 1536) */
#define MXDET_TUNE_PAR64 2        /* stride-2 dgrad: 64x128 tiles from this many tiles on (default 1600) */
#define MXDET_TUNE_WG_TARGET 3    /* grouped wgrad: workgroups per group aimed for (default 3072) */
#define MXDET_TUNE_WG_MINSTEPS 4  /* grouped wgrad: fewest 32-pixel steps per workgroup (default 64) */
#define MXDET_TUNE_WG_MAXSTEPS 5  /* grouped wgrad: most steps per workgroup (default 128) */
#define MXDET_TUNE_T3_ENABLE 6    /* wgrad: 1 = 3x3 / stride 1 / pad 1 layers use the three-tap tile (wgrad3_tile.h; default 1) */
#define MXDET_TUNE_T3_TARGET 7    /* ... workgroups of that kernel per group aimed for (default 1536) */
#define MXDET_TUNE_T3_MINSTEPS 8  /* ... fewest 64-pixel steps per workgroup (default 32: neutral for the two large groups of the single-GPU step, +1.9 % for the five smaller ones of the exchange schedule) */
#define MXDET_TUNE_T3_NS 9        /* ... LDS-DMA ring depth, 2 or 3 (default 2) */
#define MXDET_TUNE_TAIL 10        /* conv: tiles of the rows left over by the 256x256 rounds: 0 = 128x128, 1 = 64x128, 2 = 64x64 */
#define MXDET_TUNE_WG_NS 11       /* grouped wgrad (128x128 tiles): LDS-DMA ring depth 2, 3 or 4 */
#define MXDET_TUNE_ROI_TABLE 12   /* RoIAlign backward (gather): 1 = the three-kernel table form instead of the segment form */
#define MXDET_TUNE_ROI_ROWS 13    /* RoIAlign backward, segment form: rows per tile on maps with >= 64 rows (default 2) */
#define MXDET_TUNE_STATIC_TAPS 14 /* conv: 1 = stride-1 1x1 / 3x3 layers use the unrolled static-tap K loop (default 1) */
#define MXDET_TUNE_T128W 15     /* conv: stride-1 1x1 / 3x3 layers of >= 128 columns use 128x128 tiles of eight waves from this many
                                    128x128 tiles on (default 1000000 = never; measured in profiles/r02_d_static_cfg_sweep.txt) */
#define MXDET_TUNE_T3_MIX 16    /* grouped wgrad: three-tap and one-tap tiles in ONE grid (0 = two launches; 1 = one-tap ring of 3
                                    stages inside the three-tap kernel's LDS; 2 = ring of 2; default 1) */
#define MXDET_TUNE_SPLITK_TILE 17 /* mxdet_conv2d_fwd_splitk: 0 = 64x64 tiles, 1 = 128x128 (four waves), 2 = 128x128 (eight waves) */
#define MXDET_TUNE_T3_PER_ITEM 18  /* grouped wgrad, three-tap tiles: at most this many workgroups per 3x3 layer of the group (0 = no cap; default 192) */
#define MXDET_TUNE_COUNT 19
int mxdet_debug_set_tuning(int32_t which, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* MXDET_DEBUG_H_ */
