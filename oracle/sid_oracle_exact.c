/*
 * Bit-exact part of the C oracle (compiled with -ffp-contract=off): mask threshold + bilinear
 * downsample, ldm/modules/attention.py:270-284.  TEST INFRASTRUCTURE ONLY.
 */
#include <math.h>

static float thresh(float v, float ch) {
    float t = v;
    if (v < 0.5f) t = -1.0f;
    if (v > 0.5f) t = 1.0f;
    return t * ch;
}

static void taps(int dst, int in_size, float scale, int* i0, int* i1, float* w0, float* w1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    int a = (int)floorf(src);
    if (a > in_size - 1) a = in_size - 1;
    float l1 = src - (float)a;
    if (l1 < 0.f) l1 = 0.f;
    if (l1 > 1.f) l1 = 1.f;
    *i0 = a; *i1 = a + (a < in_size - 1 ? 1 : 0); *w1 = l1; *w0 = 1.0f - l1;
}

void sid_oracle_mask_downsample(const float* mask, int Hin, int Win, int Hout, int Wout, float ch, float* out) {
    const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
    for (int oy = 0; oy < Hout; ++oy)
        for (int ox = 0; ox < Wout; ++ox) {
            if (Hin == Hout && Win == Wout) { out[oy * Wout + ox] = thresh(mask[oy * Win + ox], ch); continue; }
            int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
            taps(oy, Hin, sh, &y0, &y1, &wy0, &wy1);
            taps(ox, Win, sw, &x0, &x1, &wx0, &wx1);
            const float a = thresh(mask[y0 * Win + x0], ch), b = thresh(mask[y0 * Win + x1], ch);
            const float c = thresh(mask[y1 * Win + x0], ch), d = thresh(mask[y1 * Win + x1], ch);
            const float top = wx0 * a + wx1 * b, bot = wx0 * c + wx1 * d;
            out[oy * Wout + ox] = wy0 * top + wy1 * bot;
        }
}
