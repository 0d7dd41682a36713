/*
 * lfi_oracle.h — CPU restatement of the light-field interpolation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The product path (lfinterpolator_amd/, include/lfi.h)
 * never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (ichlubna/lfInterpolator @ 2025-03-10) ships no tests, golden
 * vectors or known-answer fixtures for this path, and its implementation (CUDA surfaces + WMMA,
 * glm, two empty submodules) cannot be compiled in this pipeline.  This file therefore follows the
 * reference's source text function by function (citations below, relative to /root/reference) and is
 * cross-checked against an independent numpy restatement (oracle/lfi_oracle_np.py); neither has been
 * compared with outputs of the reference binary.
 */
#ifndef LFI_ORACLE_H
#define LFI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t x, y; } lfo_int2;
typedef struct { float x, y; } lfo_float2;

/* models of the tensor path's accumulation */
enum {
    LFO_TEN_M16 = 0,   /* fp16 accumulator re-rounded after every 16-image batch: src/kernels.cu:418-448 */
    LFO_TEN_EXACT = 1  /* exact sum over all images, one RN to fp16: what an fp32-accumulating MFMA computes */
};

/* flags shared by the blend entry points */
enum {
    LFO_ALL_FOCUS = 1u    /* per-pixel focus from a focus map: template<bool allFocus> src/kernels.cu:312,398 */
};

/* ---- fp16 helpers (cuda_fp16 semantics relied on by the reference) ---- */
uint16_t lfo_f32_to_f16(float f);          /* static_cast<half>(float) on the host: RN-even, src/interpolator.cu:219 */
uint16_t lfo_f64_to_f16(double d);         /* single RN-even rounding of a double (used by the M16/EXACT models) */
float    lfo_f16_to_f32(uint16_t h);       /* half::operator float(), exact */
uint8_t  lfo_f16_to_u8_rz(uint16_t h);     /* half → unsigned char = __half2uchar_rz: truncate, saturate, NaN→0; src/kernels.cu:393 */

/* ---- host parameterisation: src/interpolator.cu:139-246, 318-337 ---- */
int  lfo_interpret_trajectory(const char *text, int cols, int rows, float out_se[4]);     /* :318-337 */
void lfo_trajectory_point(const float se[4], int views, int i, float out_xy[2]);           /* :174-182 */
void lfo_trajectory_center(const float se[4], float out_xy[2]);                            /* :189-192 */
void lfo_weights_f32(const float view_xy[2], int cols, int rows, float effect, float *out_n); /* :156-172 */
void lfo_weight_matrix_f16(const float se[4], int cols, int rows, int views, float effect,
                           uint16_t *out_vn);                                               /* :209-224 */
void lfo_offsets(const float se[4], int cols, int rows, int width, int height, float aspect,
                 float focus, lfo_float2 *out_offsets, lfo_int2 *out_focused);             /* :226-246 */
int  lfo_focus_map_ids(const float se[4], int cols, int rows, int32_t *out_ids, int max_ids); /* :194-207 */
void lfo_block_radius(int width, int height, int32_t out_xy[2]);                           /* :139-146 */

/* ---- synthetic light field; not from the reference ---- */
uint32_t lfo_hash32(uint32_t seed, uint32_t g, uint32_t y, uint32_t x, uint32_t c);
void lfo_fill_synthetic(uint8_t *planes, int n_images, int width, int height, uint32_t seed);
void lfo_fill_synthetic_plane(uint8_t *plane, int g, int width, int height, uint32_t seed);

/* ---- device-side arithmetic: src/kernels.cu ---- */

/* Warped sample coordinate of image g for every pixel of rows [y0,y1): src/kernels.cu:72-82.
 * all_focus = 0: integer offsets; 1: (int)fmaf(focus_px, offsets[g], x) with focus_px decoded from `map`
 * (RGBA8 plane, .x channel) as fmaf(map/255, range, focus): src/kernels.cu:134-137.  Coordinates are NOT clamped. */
void lfo_warp_coords(int g, int width, int height, const lfo_int2 *focused, const lfo_float2 *offsets,
                     int all_focus, const uint8_t *map, float focus, float range,
                     int y0, int y1, lfo_int2 *out_hw);

/* STD blend: src/kernels.cu:289-343.  in: [N][H][W][4] u8, out: [V][H][W][4] u8 (only rows [y0,y1) written).
 * prequant (optional, may be NULL): [V][H][W][3] float accumulators before quantisation. */
void lfo_blend_std(const uint8_t *in, int n_images, int width, int height,
                   const lfo_int2 *focused, const lfo_float2 *offsets,
                   const uint16_t *weights_vn, int views, int v0, int v1,
                   unsigned flags, const uint8_t *map, float focus, float range,
                   int y0, int y1, uint8_t *out, float *prequant);

/* TEN_WM blend model: src/kernels.cu:345-462 with K padded to a multiple of 16 by zero weights
 * (intended math; reference defects D1/D2 of SURVEY.md §3.7 are not reproduced).
 * prequant (optional): [V][H][W][3] float = value of the final fp16 accumulator. */
void lfo_blend_ten(const uint8_t *in, int n_images, int width, int height,
                   const lfo_int2 *focused, const lfo_float2 *offsets,
                   const uint16_t *weights_vn, int views, int v0, int v1,
                   unsigned flags, const uint8_t *map, float focus, float range,
                   int model, int y0, int y1, uint8_t *out, float *prequant);

/* exact fp64 weighted mean (no rounding at all), for the ≤1e-3 normalised tolerance check */
void lfo_blend_f64(const uint8_t *in, int n_images, int width, int height,
                   const lfo_int2 *focused, const lfo_float2 *offsets,
                   const uint16_t *weights_vn, int views, int v0, int v1,
                   unsigned flags, const uint8_t *map, float focus, float range,
                   int y0, int y1, double *out_vhw3);

/* Focus map: src/kernels.cu:164-282.  map0/map1: [H][W][4] u8. */
void lfo_focus_estimate(const uint8_t *in, int n_images, int width, int height,
                        const lfo_float2 *offsets, const int32_t *ids, int n_ids,
                        float focus, float range, const int32_t block_radius[2],
                        int y0, int y1, uint8_t *map0);
void lfo_focus_filter(const uint8_t *map0, int width, int height, const int32_t block_radius[2],
                      int y0, int y1, uint8_t *map1);

#ifdef __cplusplus
}
#endif
#endif
