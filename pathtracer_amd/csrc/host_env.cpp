// HDRI importance-sampling cache, restated from the reference's CalculateHdriCache
// (Source/PathTracer/PathTracingRenderer.cpp:803-901).
//
// The table has the HDRI's own size.  Cell (row i, column j) stores
//   R = x / W  where x = lower_bound(marginal CDF over columns, i / H)
//   G = y / H  where y = lower_bound(conditional CDF of column x over rows, j / W)
//   B = normalised luminance of HDRI texel (i, j)
// (yes: the row index drives the column search and vice versa -- the shader looks the
// table up at uv = (xi1, xi2), so xi2 picks the column x and xi1 the row y).
//
// Every float accumulation below runs in the reference's order (row-major luminance sum,
// column sums over rows, serial prefix sums); compile with -ffp-contract=off.
#include <algorithm>
#include <vector>

#include "host.h"

namespace ptcore {

void CalculateHdriCache(const float* rgb, int width, int height, float* cacheOut)
{
    const size_t W = (size_t)width, H = (size_t)height;
    std::vector<float> pdf(W * H);  // [i * W + j]
    float lumSum = 0.0f;
    for (size_t i = 0; i < H; i++) {
        for (size_t j = 0; j < W; j++) {
            const float* px = rgb + 3 * (i * W + j);
            float lum = 0.212671f * px[0] + 0.715160f * px[1] + 0.072169f * px[2];
            pdf[i * W + j] = lum;
            lumSum += lum;
        }
    }
    for (size_t k = 0; k < W * H; k++) pdf[k] /= lumSum;

    std::vector<float> pdfX(W, 0.0f);
    for (size_t j = 0; j < W; j++)
        for (size_t i = 0; i < H; i++) pdfX[j] += pdf[i * W + j];

    std::vector<float> cdfX = pdfX;
    for (size_t j = 1; j < W; j++) cdfX[j] += cdfX[j - 1];

    // conditional CDF of y given x, stored per column: cdfY[j * H + i]
    std::vector<float> cdfY(W * H);
    for (size_t j = 0; j < W; j++) {
        float* col = cdfY.data() + j * H;
        for (size_t i = 0; i < H; i++) col[i] = pdf[i * W + j] / pdfX[j];
        for (size_t i = 1; i < H; i++) col[i] += col[i - 1];
    }

    for (size_t j = 0; j < W; j++) {
        const float xi2 = (float)j / (float)width;
        for (size_t i = 0; i < H; i++) {
            const float xi1 = (float)i / (float)height;
            size_t x = (size_t)(std::lower_bound(cdfX.begin(), cdfX.end(), xi1) - cdfX.begin());
            // The reference indexes cdf_y_condiciton[x] unguarded; lower_bound returns W when
            // rounding leaves the last CDF entry below xi1.  Clamp instead of reading out of bounds.
            const size_t xc = x < W ? x : W - 1;
            const float* col = cdfY.data() + xc * H;
            size_t y = (size_t)(std::lower_bound(col, col + H, xi2) - col);
            float* out = cacheOut + 3 * (i * W + j);
            out[0] = (float)x / (float)width;
            out[1] = (float)y / (float)height;
            out[2] = pdf[i * W + j];
        }
    }
}

}  // namespace ptcore
