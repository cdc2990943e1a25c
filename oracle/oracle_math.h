// TEST INFRASTRUCTURE ONLY -- never included by, linked into, or called from the product.
//
// GLSL built-ins as the oracle defines them.  The reference runs on an OpenGL driver whose
// sin/cos/asin/atan/pow/inversesqrt/inverse are vendor-defined at the ULP level (parity
// unpinned there).  The oracle therefore fixes ONE definition of each, built only from IEEE-754
// binary32 add, subtract, multiply, divide, sqrt, comparisons and integer bit operations,
// each rounded once, in the order written (compile with -ffp-contract=off, no fast-math).
// Any platform that executes the same sequence gets the same bits.  The transcendental
// approximations are the classic single-precision Cody-Waite / minimax schemes
// (1-2 ulp over the ranges the shader uses).
//
// GLSL semantics restated (GLSL 4.60 spec section 8):
//   min(x,y)/max(x,y): undefined for NaN in GLSL; defined here as IEEE minimumNumber/maximumNumber.
//   mix(x,y,a) = x*(1-a) + y*a        clamp(x,lo,hi) = min(max(x,lo),hi)
//   inversesqrt(x) = 1/sqrt(x)        normalize(v) = v * inversesqrt(dot(v,v))
//   reflect(I,N) = I - 2*dot(N,I)*N   refract: k = 1 - eta*eta*(1 - dot(N,I)^2)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

inline float bits2f(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t f2bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// IEEE 754-2019 minimumNumber / maximumNumber: a NaN operand is ignored, and -0 orders below +0.
inline float gl_min(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2bits(a) >> 31) ? a : b;
    return (b < a) ? b : a;
}
inline float gl_max(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2bits(a) >> 31) ? b : a;
    return (a < b) ? b : a;
}
inline float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
inline float gl_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline float gl_abs(float x) { return bits2f(f2bits(x) & 0x7fffffffu); }
inline float gl_sqrt(float x) { return std::sqrt(x); }  // IEEE correctly rounded
inline float gl_inversesqrt(float x) { return 1.0f / std::sqrt(x); }

// ---- sin / cos: octant reduction by pi/4 in three exact-product steps, degree-7/8 minimax.
constexpr float kFOPI = 1.27323954473516f;
constexpr float kDP1 = 0.78515625f;
constexpr float kDP2 = 2.4187564849853515625e-4f;
constexpr float kDP3 = 3.77489497744594108e-8f;

inline float poly_sin(float x, float z) { return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z + -1.6666654611e-1f) * z * x + x; }
inline float poly_cos(float z)
{
    float y = ((2.443315711809948e-5f * z + -1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}

inline float gl_sin(float xx)
{
    bool neg = xx < 0.0f;
    float x = gl_abs(xx);
    int j = (int)(kFOPI * x);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    if (j > 3) {
        neg = !neg;
        j -= 4;
    }
    x = ((x - y * kDP1) - y * kDP2) - y * kDP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? poly_cos(z) : poly_sin(x, z);
    return neg ? -r : r;
}

inline float gl_cos(float xx)
{
    bool neg = false;
    float x = gl_abs(xx);
    int j = (int)(kFOPI * x);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    if (j > 3) {
        neg = !neg;
        j -= 4;
    }
    if (j > 1) neg = !neg;
    x = ((x - y * kDP1) - y * kDP2) - y * kDP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? poly_sin(x, z) : poly_cos(z);
    return neg ? -r : r;
}

// ---- asin: input clamped to [-1, 1] (GLSL leaves |x| > 1 undefined).
constexpr float kPIO2 = 1.5707963267948966192f;
constexpr float kPIO4 = 0.7853981633974483096f;
constexpr float kPI = 3.14159265358979323846f;

inline float gl_asin(float xx)
{
    bool neg = xx < 0.0f;
    float a = gl_abs(xx);
    if (a > 1.0f) a = 1.0f;
    float x, z;
    bool big = a > 0.5f;
    if (big) {
        z = 0.5f * (1.0f - a);
        x = gl_sqrt(z);
    } else {
        x = a;
        z = x * x;
    }
    float r;
    if (a < 1.0e-4f) {
        r = a;
    } else {
        r = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
             1.6666752422e-1f) * z * x + x;
        if (big) {
            r = r + r;
            r = kPIO2 - r;
        }
    }
    return neg ? -r : r;
}

inline float gl_atan1(float xx)
{
    bool neg = xx < 0.0f;
    float x = gl_abs(xx);
    float y;
    if (x > 2.414213562373095f) {
        y = kPIO2;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = kPIO4;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return neg ? -y : y;
}

// GLSL atan(y, x).  atan(0, 0) is undefined in GLSL; defined here as 0.
inline float gl_atan2(float y, float x)
{
    if (x > 0.0f) return gl_atan1(y / x);
    if (x < 0.0f) return (y >= 0.0f) ? gl_atan1(y / x) + kPI : gl_atan1(y / x) - kPI;
    if (y > 0.0f) return kPIO2;
    if (y < 0.0f) return -kPIO2;
    return 0.0f;
}

// ---- pow(x, y) = exp(y * log(x)) for x > 0; pow(0, y > 0) = 0; x < 0 undefined in GLSL (NaN here).
inline float scale_pow2(float v, int n)  // v * 2^n without libm
{
    if (n > 127) {
        v = v * bits2f(254u << 23);
        n -= 127;
        if (n > 127) n = 127;
    } else if (n < -126) {
        v = v * bits2f(1u << 23);
        n += 126;
        if (n < -126) n = -126;
    }
    return v * bits2f((uint32_t)(n + 127) << 23);
}

inline float gl_log(float xx)  // natural log, x > 0 finite
{
    float x = xx;
    int e = 0;
    uint32_t u = f2bits(x);
    if ((u >> 23) == 0) {  // subnormal: renormalise
        x = x * 16777216.0f;
        u = f2bits(x);
        e = -24;
    }
    e += (int)(u >> 23) - 126;
    x = bits2f((u & 0x007fffffu) | 0x3f000000u);  // mantissa in [0.5, 1)
    if (x < 0.707106781186547524f) {
        e -= 1;
        x = x + x - 1.0f;
    } else {
        x = x - 1.0f;
    }
    float z = x * x;
    float y = ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x - 1.2420140846e-1f) * x +
                   1.4249322787e-1f) * x - 1.6668057665e-1f) * x + 2.0000714765e-1f) * x - 2.4999993993e-1f) * x +
               3.3333331174e-1f) * x * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = x + y;
    z = z + 0.693359375f * fe;
    return z;
}

inline float gl_exp(float xx)
{
    if (xx > 88.72283905206835f) return bits2f(0x7f800000u);
    if (xx < -103.278929903431851103f) return 0.0f;
    float x = xx;
    float z = std::floor(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
          1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return scale_pow2(z, n);
}

inline float gl_pow(float x, float y)
{
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : bits2f(0x7fc00000u);
    if (x < 0.0f || x != x) return bits2f(0x7fc00000u);
    return gl_exp(y * gl_log(x));
}

}  // namespace oracle
