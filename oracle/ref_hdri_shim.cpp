// TEST INFRASTRUCTURE ONLY -- never linked into, or called from, the product library.
//
// extern "C" wrapper around the arithmetic of the reference's HDRI-cache builder,
// PathTracingRenderer::CalculateHdriCache (Source/PathTracer/PathTracingRenderer.cpp:776-919).
// The function as a whole cannot be compiled here (it reads its input from a GL texture object and
// returns one), but lines 803-901 of that file are plain standard C++ over a std::vector<float>
// named `textureData` and two ints `width`, `height`, and leave the result in `float* cache`.
// oracle/Makefile extracts exactly those lines from the reference file, where it lies, into
// oracle/_ref/hdri_cache_803_901.inc at build time (git-ignored, never committed) and this file
// includes them unmodified.  No reference source text lives in this file.
//
// The one dialect adaptation is on the compiler command line: `-D__int64="long long"` (lines 878 and
// 881 use MSVC's built-in 64-bit integer keyword).
#include <algorithm>
#include <cstring>
#include <vector>

extern "C" void ref_calculate_hdri_cache(const float* hdri_rgb, int width, int height, float* cache_out)
{
    std::vector<float> textureData(hdri_rgb, hdri_rgb + (size_t)width * height * 3);
#include "hdri_cache_803_901.inc"
    std::memcpy(cache_out, cache, sizeof(float) * (size_t)width * height * 3);
    delete[] cache;
}
