/*
 * lfi_oracle.c — scalar CPU restatement of the reference's shift-and-sum path.
 *
 * TEST INFRASTRUCTURE ONLY (see lfi_oracle.h): checker for tests/, smoke() and the cpu_baseline leg of
 * bench.py.  PARITY UNPINNED: no golden vectors exist in the reference and its CUDA code cannot be run
 * here; every function cites the reference lines (relative to /root/reference) whose arithmetic it restates.
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off: explicit fmaf only, nothing contracted).
 */
#include "lfi_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * fp16 helpers
 * ---------------------------------------------------------------------------------------------- */

uint16_t lfo_f64_to_f16(double d)
{
    uint16_t sign = signbit(d) ? 0x8000u : 0u;
    double a = fabs(d);
    if(isnan(d))
        return (uint16_t)(sign | 0x7e00u);
    if(a >= 65520.0) /* halfway between the largest finite half (65504) and 2^16 rounds up to infinity */
        return (uint16_t)(sign | 0x7c00u);
    if(a == 0.0)
        return sign;
    int ex;
    (void)frexp(a, &ex);
    int e = ex - 1; /* a in [2^e, 2^(e+1)) */
    if(e < -14)
    {
        /* subnormal half: multiples of 2^-24 */
        double q = rint(ldexp(a, 24)); /* RN-even in the default rounding mode */
        return (uint16_t)(sign | (uint16_t)q); /* q == 1024 is the smallest normal, same bit pattern */
    }
    double q = rint(ldexp(a, 10 - e)); /* in [1024, 2048] */
    if(q >= 2048.0)
    {
        q = 1024.0;
        e++;
    }
    return (uint16_t)(sign | (uint16_t)(((e + 15) << 10) | ((int)q - 1024)));
}

uint16_t lfo_f32_to_f16(float f)
{
    /* float → double is exact, so one rounding happens */
    return lfo_f64_to_f16((double)f);
}

float lfo_f16_to_f32(uint16_t h)
{
    int sign = h >> 15;
    int e = (h >> 10) & 0x1f;
    int m = h & 0x3ff;
    double v;
    if(e == 0)
        v = ldexp((double)m, -24);
    else if(e == 31)
        v = m ? (double)NAN : (double)INFINITY;
    else
        v = ldexp((double)(m | 0x400), e - 25);
    return (float)(sign ? -v : v);
}

uint8_t lfo_f16_to_u8_rz(uint16_t h)
{
    float v = lfo_f16_to_f32(h);
    if(isnan(v) || v <= 0.0f)
        return 0;
    if(v >= 255.0f)
        return 255;
    return (uint8_t)v; /* C conversion truncates toward zero */
}

/* ------------------------------------------------------------------------------------------------
 * Host parameterisation
 * ---------------------------------------------------------------------------------------------- */

/* glm::distance for vec2 = sqrt(dx*dx + dy*dy) in float (src/interpolator.cu:158,164,200) */
static float dist2f(float ax, float ay, float bx, float by)
{
    float dx = bx - ax, dy = by - ay;
    float xx = dx * dx, yy = dy * dy;
    return sqrtf(xx + yy);
}

/* src/interpolator.cu:318-337: comma separated floats, value_i * (colsRows[i % 2] - 1) */
int lfo_interpret_trajectory(const char *text, int cols, int rows, float out_se[4])
{
    int dims[2] = {cols, rows};
    int count = 0;
    const char *p = text;
    for(int i = 0; i < 4; i++)
        out_se[i] = 0.0f;
    while(*p && count < 4)
    {
        char *end = NULL;
        float value = strtof(p, &end);
        if(end == p)
            return -1;
        out_se[count] = value * (float)(dims[count % 2] - 1);
        count++;
        p = end;
        while(*p && *p != ',')
            p++;
        if(*p == ',')
            p++;
    }
    return count;
}

/* src/interpolator.cu:174-182: start + step*i with step = (end-start)/(views-1); a single view sits on the start */
void lfo_trajectory_point(const float se[4], int views, int i, float out_xy[2])
{
    if(views <= 1)
    {
        out_xy[0] = se[0];
        out_xy[1] = se[1];
        return;
    }
    float denom = (float)(views - 1);
    float sx = (se[2] - se[0]) / denom;
    float sy = (se[3] - se[1]) / denom;
    float px = sx * (float)i;
    float py = sy * (float)i;
    out_xy[0] = se[0] + px;
    out_xy[1] = se[1] + py;
}

/* src/interpolator.cu:189-192 */
void lfo_trajectory_center(const float se[4], float out_xy[2])
{
    float hx = (se[2] - se[0]) * 0.5f;
    float hy = (se[3] - se[1]) * 0.5f;
    out_xy[0] = se[0] + hx;
    out_xy[1] = se[1] + hy;
}

/* src/interpolator.cu:156-172: inverse-distance power law, normalised by a sequential float sum, g = col*rows + row */
void lfo_weights_f32(const float view_xy[2], int cols, int rows, float effect, float *out_n)
{
    float max_distance = dist2f(0.0f, 0.0f, (float)cols, (float)rows);
    float sum = 0.0f;
    int g = 0;
    for(int col = 0; col < cols; col++)
        for(int row = 0; row < rows; row++)
        {
            float w = max_distance - dist2f(view_xy[0], view_xy[1], (float)col, (float)row);
            w = powf(w, effect);
            sum += w;
            out_n[g++] = w;
        }
    for(int i = 0; i < g; i++)
        out_n[i] /= sum;
}

/* src/interpolator.cu:209-224: one row of fp16 weights per trajectory view */
void lfo_weight_matrix_f16(const float se[4], int cols, int rows, int views, float effect, uint16_t *out_vn)
{
    int n = cols * rows;
    float *line = (float *)malloc(sizeof(float) * (size_t)n);
    for(int v = 0; v < views; v++)
    {
        float xy[2];
        lfo_trajectory_point(se, views, v, xy);
        lfo_weights_f32(xy, cols, rows, effect, line);
        for(int g = 0; g < n; g++)
            out_vn[(size_t)v * n + g] = lfo_f32_to_f16(line[g]);
    }
    free(line);
}

/* src/interpolator.cu:226-246 */
void lfo_offsets(const float se[4], int cols, int rows, int width, int height, float aspect, float focus,
                 lfo_float2 *out_offsets, lfo_int2 *out_focused)
{
    float center[2];
    lfo_trajectory_center(se, center);
    float offset_aspect = ((float)width / (float)height) / aspect;
    int g = 0;
    for(int col = 0; col < cols; col++)
        for(int row = 0; row < rows; row++)
        {
            float ox = (center[0] - (float)col) / (float)cols;
            float oy = (center[1] - (float)row) / (float)rows;
            ox *= (float)width;
            oy *= (float)height;
            oy *= offset_aspect;
            out_offsets[g].x = ox;
            out_offsets[g].y = oy;
            float fx = ox * focus, fy = oy * focus;
            out_focused[g].x = (int32_t)roundf(fx); /* glm::round: half away from zero */
            out_focused[g].y = (int32_t)roundf(fy);
            g++;
        }
}

typedef struct { float d; int32_t id; } dist_id;

static int cmp_dist_id(const void *a, const void *b)
{
    const dist_id *p = (const dist_id *)a, *q = (const dist_id *)b;
    if(p->d < q->d) return -1;
    if(p->d > q->d) return 1;
    return (p->id > q->id) - (p->id < q->id); /* ties: the reference's std::sort leaves them unspecified; fixed by id */
}

/* src/interpolator.cu:194-207: the (up to) 32 grid images nearest to the trajectory centre */
int lfo_focus_map_ids(const float se[4], int cols, int rows, int32_t *out_ids, int max_ids)
{
    int n = cols * rows;
    float center[2];
    lfo_trajectory_center(se, center);
    dist_id *all = (dist_id *)malloc(sizeof(dist_id) * (size_t)n);
    int g = 0;
    for(int col = 0; col < cols; col++)
        for(int row = 0; row < rows; row++)
        {
            all[g].d = dist2f((float)col, (float)row, center[0], center[1]);
            all[g].id = g;
            g++;
        }
    qsort(all, (size_t)n, sizeof(dist_id), cmp_dist_id);
    int count = n < max_ids ? n : max_ids;
    for(int i = 0; i < count; i++)
        out_ids[i] = all[i].id;
    free(all);
    return count;
}

/* src/interpolator.cu:139-146: resolution/100 bumped to even; 0 would never advance the tap loops (D6) so it becomes 1 */
void lfo_block_radius(int width, int height, int32_t out_xy[2])
{
    int32_t r[2] = {width / 100, height / 100};
    for(int i = 0; i < 2; i++)
    {
        if(r[i] % 2 != 0)
            r[i]++;
        if(r[i] < 1)
            r[i] = 1;
        out_xy[i] = r[i];
    }
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic light field
 * ---------------------------------------------------------------------------------------------- */

static uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

uint32_t lfo_hash32(uint32_t seed, uint32_t g, uint32_t y, uint32_t x, uint32_t c)
{
    uint32_t h = mix32(seed + g * 0x9e3779b9u);
    h = mix32(h + y * 0x85ebca6bu);
    h = mix32(h + x * 0xc2b2ae35u + c);
    return h;
}

void lfo_fill_synthetic_plane(uint8_t *plane, int g, int width, int height, uint32_t seed)
{
    for(int y = 0; y < height; y++)
    {
        uint8_t *row = plane + (size_t)y * (size_t)width * 4;
        for(int x = 0; x < width; x++)
        {
            for(int c = 0; c < 3; c++)
                row[4 * x + c] = (uint8_t)(lfo_hash32(seed, (uint32_t)g, (uint32_t)y, (uint32_t)x, (uint32_t)c) >> 24);
            row[4 * x + 3] = 255;
        }
    }
}

void lfo_fill_synthetic(uint8_t *planes, int n_images, int width, int height, uint32_t seed)
{
    for(int g = 0; g < n_images; g++)
        lfo_fill_synthetic_plane(planes + (size_t)g * height * (size_t)width * 4, g, width, height, seed);
}

/* ------------------------------------------------------------------------------------------------
 * Warp + fetch
 * ---------------------------------------------------------------------------------------------- */

static int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : (v > hi ? hi : v);
}

/* surf2Dread(..., cudaBoundaryModeClamp): src/kernels.cu:119-126 */
static const uint8_t *fetch_px(const uint8_t *plane, int width, int height, int x, int y)
{
    x = clampi(x, 0, width - 1);
    y = clampi(y, 0, height - 1);
    return plane + ((size_t)y * width + x) * 4;
}

/* src/kernels.cu:134-137; nvcc contracts a + b*c into one fma */
static float decode_focus(uint8_t map_value, float focus, float range)
{
    float t = (float)map_value / 255.0f;
    return fmaf(t, range, focus);
}

/* src/kernels.cu:72-76 (integer offsets) and :78-82 (float offsets, C truncation, contracted fma) */
static void warp_px(int x, int y, int g, const lfo_int2 *focused, const lfo_float2 *offsets, int all_focus,
                    float focus_px, int *ox, int *oy)
{
    if(all_focus)
    {
        *ox = (int)fmaf(focus_px, offsets[g].x, (float)x);
        *oy = (int)fmaf(focus_px, offsets[g].y, (float)y);
    }
    else
    {
        *ox = x + focused[g].x;
        *oy = y + focused[g].y;
    }
}

void lfo_warp_coords(int g, int width, int height, const lfo_int2 *focused, const lfo_float2 *offsets,
                     int all_focus, const uint8_t *map, float focus, float range, int y0, int y1, lfo_int2 *out_hw)
{
    for(int y = y0; y < y1; y++)
        for(int x = 0; x < width; x++)
        {
            float fpx = 0.0f;
            if(all_focus)
                fpx = decode_focus(fetch_px(map, width, height, x, y)[0], focus, range);
            int sx, sy;
            warp_px(x, y, g, focused, offsets, all_focus, fpx, &sx, &sy);
            out_hw[(size_t)y * width + x].x = sx;
            out_hw[(size_t)y * width + x].y = sy;
        }
}

/* ------------------------------------------------------------------------------------------------
 * STD blend
 * ---------------------------------------------------------------------------------------------- */

void lfo_blend_std(const uint8_t *in, int n_images, int width, int height, const lfo_int2 *focused,
                   const lfo_float2 *offsets, const uint16_t *weights_vn, int views, int v0, int v1, unsigned flags,
                   const uint8_t *map, float focus, float range, int y0, int y1, uint8_t *out, float *prequant)
{
    const int all_focus = (flags & LFO_ALL_FOCUS) != 0;
    const int nv = v1 - v0;
    const size_t plane = (size_t)width * height * 4;
    (void)views;
    /* weights as float, transposed to [g][v] so the view loop is contiguous; half → float is exact */
    float *wt = (float *)malloc(sizeof(float) * (size_t)n_images * nv);
    for(int g = 0; g < n_images; g++)
        for(int v = 0; v < nv; v++)
            wt[(size_t)g * nv + v] = lfo_f16_to_f32(weights_vn[(size_t)(v0 + v) * n_images + g]);
    float *sum = (float *)malloc(sizeof(float) * 3 * (size_t)nv);

    for(int y = y0; y < y1; y++)
        for(int x = 0; x < width; x++)
        {
            for(int i = 0; i < 3 * nv; i++)
                sum[i] = 0.0f;
            float fpx = 0.0f;
            if(all_focus)
                fpx = decode_focus(fetch_px(map, width, height, x, y)[0], focus, range);
            for(int g = 0; g < n_images; g++) /* ascending g: src/kernels.cu:328 */
            {
                int sx, sy;
                warp_px(x, y, g, focused, offsets, all_focus, fpx, &sx, &sy);
                const uint8_t *px = fetch_px(in + plane * g, width, height, sx, sy);
                const float r = (float)px[0], gr = (float)px[1], b = (float)px[2];
                const float *w = wt + (size_t)g * nv;
                for(int v = 0; v < nv; v++) /* addWeighted: src/kernels.cu:292-299 */
                {
                    sum[3 * v + 0] = fmaf(r, w[v], sum[3 * v + 0]);
                    sum[3 * v + 1] = fmaf(gr, w[v], sum[3 * v + 1]);
                    sum[3 * v + 2] = fmaf(b, w[v], sum[3 * v + 2]);
                }
            }
            for(int v = 0; v < nv; v++) /* uch4: __float2int_rn then narrowing to unsigned char, src/kernels.cu:301-310 */
            {
                uint8_t *o = out + plane * (size_t)(v0 + v) + ((size_t)y * width + x) * 4;
                for(int c = 0; c < 3; c++)
                {
                    float s = sum[3 * v + c];
                    o[c] = (uint8_t)(int)rintf(s);
                    if(prequant)
                        prequant[(((size_t)(v0 + v) * height + y) * width + x) * 3 + c] = s;
                }
                o[3] = 255;
            }
        }
    free(sum);
    free(wt);
}

/* ------------------------------------------------------------------------------------------------
 * TEN_WM blend models
 * ---------------------------------------------------------------------------------------------- */

void lfo_blend_ten(const uint8_t *in, int n_images, int width, int height, const lfo_int2 *focused,
                   const lfo_float2 *offsets, const uint16_t *weights_vn, int views, int v0, int v1, unsigned flags,
                   const uint8_t *map, float focus, float range, int model, int y0, int y1, uint8_t *out,
                   float *prequant)
{
    const int all_focus = (flags & LFO_ALL_FOCUS) != 0;
    const int nv = v1 - v0;
    const int batch = 16;                                   /* IMAGES: src/kernels.cu:350 */
    const int n_pad = (n_images + batch - 1) / batch * batch; /* zero-weight padding instead of dropping the tail (D2) */
    const size_t plane = (size_t)width * height * 4;
    (void)views;
    double *wt = (double *)calloc((size_t)n_pad * nv, sizeof(double)); /* [g][v], exact */
    for(int g = 0; g < n_images; g++)
        for(int v = 0; v < nv; v++)
            wt[(size_t)g * nv + v] = (double)lfo_f16_to_f32(weights_vn[(size_t)(v0 + v) * n_images + g]);
    uint8_t *pix = (uint8_t *)calloc((size_t)n_pad * 3, 1);

    for(int y = y0; y < y1; y++)
        for(int x = 0; x < width; x++)
        {
            float fpx = 0.0f;
            if(all_focus)
                fpx = decode_focus(fetch_px(map, width, height, x, y)[0], focus, range);
            for(int g = 0; g < n_images; g++) /* loadPixels: src/kernels.cu:353-370 */
            {
                int sx, sy;
                warp_px(x, y, g, focused, offsets, all_focus, fpx, &sx, &sy);
                const uint8_t *px = fetch_px(in + plane * g, width, height, sx, sy);
                pix[3 * g + 0] = px[0];
                pix[3 * g + 1] = px[1];
                pix[3 * g + 2] = px[2];
            }
            for(int v = 0; v < nv; v++)
            {
                uint8_t *o = out + plane * (size_t)(v0 + v) + ((size_t)y * width + x) * 4;
                for(int c = 0; c < 3; c++)
                {
                    uint16_t acc = 0; /* fill_fragment(0): src/kernels.cu:423-425 */
                    double total = 0.0;
                    for(int b0 = 0; b0 < n_pad; b0 += batch)
                    {
                        /* products of an fp16 weight and an 8-bit integer and their sums are exact in double */
                        double s = 0.0;
                        for(int k = 0; k < batch; k++)
                            s += wt[(size_t)(b0 + k) * nv + v] * (double)pix[3 * (b0 + k) + c];
                        total += s;
                        if(model == LFO_TEN_M16) /* mma_sync with a half accumulator fragment: src/kernels.cu:422,446 */
                            acc = lfo_f64_to_f16((double)lfo_f16_to_f32(acc) + s);
                    }
                    if(model != LFO_TEN_M16)
                        acc = lfo_f64_to_f16(total);
                    o[c] = lfo_f16_to_u8_rz(acc); /* storePortionViews: src/kernels.cu:387-396 */
                    if(prequant)
                        prequant[(((size_t)(v0 + v) * height + y) * width + x) * 3 + c] = lfo_f16_to_f32(acc);
                }
                o[3] = 255;
            }
        }
    free(pix);
    free(wt);
}

void lfo_blend_f64(const uint8_t *in, int n_images, int width, int height, const lfo_int2 *focused,
                   const lfo_float2 *offsets, const uint16_t *weights_vn, int views, int v0, int v1, unsigned flags,
                   const uint8_t *map, float focus, float range, int y0, int y1, double *out_vhw3)
{
    const int all_focus = (flags & LFO_ALL_FOCUS) != 0;
    const int nv = v1 - v0;
    const size_t plane = (size_t)width * height * 4;
    (void)views;
    for(int y = y0; y < y1; y++)
        for(int x = 0; x < width; x++)
        {
            float fpx = 0.0f;
            if(all_focus)
                fpx = decode_focus(fetch_px(map, width, height, x, y)[0], focus, range);
            for(int v = 0; v < nv; v++)
                for(int c = 0; c < 3; c++)
                    out_vhw3[(((size_t)(v0 + v) * height + y) * width + x) * 3 + c] = 0.0;
            for(int g = 0; g < n_images; g++)
            {
                int sx, sy;
                warp_px(x, y, g, focused, offsets, all_focus, fpx, &sx, &sy);
                const uint8_t *px = fetch_px(in + plane * g, width, height, sx, sy);
                for(int v = 0; v < nv; v++)
                {
                    double w = (double)lfo_f16_to_f32(weights_vn[(size_t)(v0 + v) * n_images + g]);
                    double *o = out_vhw3 + (((size_t)(v0 + v) * height + y) * width + x) * 3;
                    o[0] += w * px[0];
                    o[1] += w * px[1];
                    o[2] += w * px[2];
                }
            }
        }
}

/* ------------------------------------------------------------------------------------------------
 * Focus map
 * ---------------------------------------------------------------------------------------------- */

/* focusDispersion: src/kernels.cu:196-217 with ElementRange :173-194 and distance :167-170 */
static float focus_dispersion(const uint8_t *in, int width, int height, const lfo_float2 *offsets, const int32_t *ids,
                              int n_ids, const int32_t radius[2], float f, int x, int y)
{
    const size_t plane = (size_t)width * height * 4;
    float lo[9][3], hi[9][3];
    for(int i = 0; i < 9; i++)
        for(int c = 0; c < 3; c++)
        {
            lo[i][c] = FLT_MAX;
            hi[i][c] = FLT_MIN; /* sic: smallest positive float, as in the reference (:178) */
        }
    for(int k = 0; k < n_ids; k++)
    {
        int g = ids[k];
        int cx = (int)fmaf(f, offsets[g].x, (float)x);
        int cy = (int)fmaf(f, offsets[g].y, (float)y);
        int i = 0;
        for(int tx = cx - radius[0]; tx <= cx + radius[0]; tx += radius[0])
            for(int ty = cy - radius[1]; ty <= cy + radius[1]; ty += radius[1])
            {
                const uint8_t *px = fetch_px(in + plane * g, width, height, tx, ty);
                for(int c = 0; c < 3; c++)
                {
                    lo[i][c] = fminf(lo[i][c], (float)px[c]);
                    hi[i][c] = fmaxf(hi[i][c], (float)px[c]);
                }
                i++;
            }
    }
    float total = 0.0f;
    for(int i = 0; i < 9; i++)
        total += fmaxf(fmaxf(fabsf(lo[i][0] - hi[i][0]), fabsf(lo[i][1] - hi[i][1])), fabsf(lo[i][2] - hi[i][2]));
    return total;
}

/* FocusMap::estimate: src/kernels.cu:239-258 */
void lfo_focus_estimate(const uint8_t *in, int n_images, int width, int height, const lfo_float2 *offsets,
                        const int32_t *ids, int n_ids, float focus, float range, const int32_t block_radius[2], int y0,
                        int y1, uint8_t *map0)
{
    const int steps = 32;
    (void)n_images;
    float step = range / (float)(steps - 1);
    for(int y = y0; y < y1; y++)
        for(int x = 0; x < width; x++)
        {
            float best_d = FLT_MAX, best_f = 0.0f; /* MinDispersion: src/kernels.cu:219-237 */
            for(int i = 0; i < steps; i++)
            {
                float f = fmaf(step, (float)i, focus);
                float d = focus_dispersion(in, width, height, offsets, ids, n_ids, block_radius, f, x, y);
                if(d < best_d)
                {
                    best_d = d;
                    best_f = f;
                }
            }
            float normalized = (best_f - focus) / range;
            uint8_t m = (uint8_t)roundf(normalized * 255.0f);
            uint8_t *o = map0 + ((size_t)y * width + x) * 4;
            o[0] = o[1] = o[2] = m;
            o[3] = 255;
        }
}

/* FocusMap::filter: src/kernels.cu:260-280; radius/10 of 0 (blockRadius < 10) would divide 0 by 0 (D6) so it becomes 1 */
void lfo_focus_filter(const uint8_t *map0, int width, int height, const int32_t block_radius[2], int y0, int y1,
                      uint8_t *map1)
{
    int rx = block_radius[0] / 10, ry = block_radius[1] / 10;
    if(rx < 1) rx = 1;
    if(ry < 1) ry = 1;
    for(int y = y0; y < y1; y++)
        for(int x = 0; x < width; x++)
        {
            float avg = 0.0f;
            int count = 0;
            for(int tx = x - rx; tx < x + rx; tx++)
                for(int ty = y - ry; ty < y + ry; ty++)
                {
                    avg += (float)fetch_px(map0, width, height, tx, ty)[0];
                    count++;
                }
            avg /= (float)count;
            uint8_t m = (uint8_t)roundf(avg);
            uint8_t *o = map1 + ((size_t)y * width + x) * 4;
            o[0] = o[1] = o[2] = m;
            o[3] = 255;
        }
}
