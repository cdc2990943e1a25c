// Device-side float32 building blocks.
//
// Parity contract: every function here is a fixed sequence of IEEE-754 binary32 operations
// (add, subtract, multiply, divide, sqrt, compare, integer bit ops), each rounded once, in
// the order written.  The translation unit is compiled with -ffp-contract=off and hipcc's
// default correctly-rounded divide/sqrt, so a CPU that executes the same sequence produces
// the same bits -- which is how the renderer is checked against its CPU oracle bit for bit.
// Transcendentals (sin, cos, asin, atan, exp, log) are therefore implemented here with
// octant / interval reduction and minimax polynomials instead of the ocml library calls,
// whose results are not reproducible off-device.
#pragma once
#include <hip/hip_runtime.h>

namespace ptcore {

#define PT_DEV __device__ __forceinline__

struct v2 {
    float x, y;
};
struct v3 {
    float x, y, z;
};

PT_DEV v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
PT_DEV v3 splat(float s) { return v3{s, s, s}; }
PT_DEV v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
PT_DEV v3 operator*(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV v3 operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
PT_DEV v3 operator*(float s, v3 a) { return v3{s * a.x, s * a.y, s * a.z}; }
PT_DEV v3 operator/(v3 a, float s) { return v3{a.x / s, a.y / s, a.z / s}; }
PT_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV v3 cross(v3 x, v3 y) { return v3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }

PT_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
PT_DEV float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
PT_DEV float mix_(float x, float y, float a) { return x * (1.0f - a) + y * a; }
PT_DEV v3 mix3(v3 x, v3 y, float a) { return x * (1.0f - a) + y * a; }
PT_DEV float abs_(float x) { return __builtin_fabsf(x); }
PT_DEV float sqrt_(float x) { return __builtin_sqrtf(x); }
PT_DEV float rsqrt_(float x) { return 1.0f / __builtin_sqrtf(x); }
PT_DEV float length3(v3 a) { return sqrt_(dot(a, a)); }
PT_DEV v3 normalize3(v3 a) { return a * rsqrt_(dot(a, a)); }

// ---- sin / cos on octants of pi/4
PT_DEV float polySin(float x, float z)
{
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z + -1.6666654611e-1f) * z * x + x;
}
PT_DEV float polyCos(float z)
{
    float y = ((2.443315711809948e-5f * z + -1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    y = y - 0.5f * z;
    return y + 1.0f;
}
// Shared reduction: returns the reduced argument and its square, octant in j (0..3 after folding).
PT_DEV void reduceOctant(float ax, float& x, float& z, int& j, bool& flip)
{
    int q = (int)(1.27323954473516f * ax);
    float y = (float)q;
    if (q & 1) {
        q += 1;
        y += 1.0f;
    }
    q &= 7;
    flip = q > 3;
    j = flip ? q - 4 : q;
    x = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    z = x * x;
}
PT_DEV float sin_(float xx)
{
    float x, z;
    int j;
    bool flip;
    reduceOctant(abs_(xx), x, z, j, flip);
    bool neg = (xx < 0.0f) != flip;
    float r = (j == 1 || j == 2) ? polyCos(z) : polySin(x, z);
    return neg ? -r : r;
}
PT_DEV float cos_(float xx)
{
    float x, z;
    int j;
    bool flip;
    reduceOctant(abs_(xx), x, z, j, flip);
    bool neg = flip != (j > 1);
    float r = (j == 1 || j == 2) ? polySin(x, z) : polyCos(z);
    return neg ? -r : r;
}
// sin and cos of the same angle share the reduction.
PT_DEV void sincos_(float xx, float& s, float& c)
{
    float x, z;
    int j;
    bool flip;
    reduceOctant(abs_(xx), x, z, j, flip);
    float ps = polySin(x, z), pc = polyCos(z);
    bool swap = (j == 1 || j == 2);
    float rs = swap ? pc : ps, rc = swap ? ps : pc;
    bool negS = (xx < 0.0f) != flip;
    bool negC = flip != (j > 1);
    s = negS ? -rs : rs;
    c = negC ? -rc : rc;
}

PT_DEV float asin_(float xx)
{
    bool neg = xx < 0.0f;
    float a = abs_(xx);
    if (a > 1.0f) a = 1.0f;
    bool big = a > 0.5f;
    float x, z;
    if (big) {
        z = 0.5f * (1.0f - a);
        x = sqrt_(z);
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
            r = 1.5707963267948966192f - r;
        }
    }
    return neg ? -r : r;
}

// (One division serves both reduced ranges: -(1 / x) is (-1) / x bit for bit, rounding being symmetric in sign;
// lanes of a wave that fall into different ranges then share one copy of the division's code.)
PT_DEV float atan1_(float xx)
{
    bool neg = xx < 0.0f;
    float x = abs_(xx);
    const bool high = x > 2.414213562373095f, mid = x > 0.4142135623730950f;
    float y = high ? 1.5707963267948966192f : (mid ? 0.7853981633974483096f : 0.0f);
    if (mid) {  // (`high` implies `mid`)
        const float num = high ? -1.0f : x - 1.0f;
        const float den = high ? x : x + 1.0f;
        x = num / den;
    }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return neg ? -y : y;
}
// (The quotient and its arctangent are taken once for every x != 0 and the quadrant's offset added afterwards: the
// same operations per lane as a branch per quadrant, in one copy of the code.)
PT_DEV float atan2_(float y, float x)
{
    const float pi = 3.14159265358979323846f;
    if (!(x > 0.0f) && !(x < 0.0f)) {  // x is zero (or a NaN)
        if (y > 0.0f) return 1.5707963267948966192f;
        if (y < 0.0f) return -1.5707963267948966192f;
        return 0.0f;
    }
    const float t = atan1_(y / x);
    if (x > 0.0f) return t;
    return (y >= 0.0f) ? t + pi : t - pi;
}

PT_DEV float scalePow2(float v, int n)
{
    if (n > 127) {
        v = v * __uint_as_float(254u << 23);
        n -= 127;
        if (n > 127) n = 127;
    } else if (n < -126) {
        v = v * __uint_as_float(1u << 23);
        n += 126;
        if (n < -126) n = -126;
    }
    return v * __uint_as_float((unsigned)(n + 127) << 23);
}
PT_DEV float log_(float xx)
{
    float x = xx;
    int e = 0;
    unsigned u = __float_as_uint(x);
    if ((u >> 23) == 0) {
        x = x * 16777216.0f;
        u = __float_as_uint(x);
        e = -24;
    }
    e += (int)(u >> 23) - 126;
    x = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
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
PT_DEV float exp_(float xx)
{
    if (xx > 88.72283905206835f) return __uint_as_float(0x7f800000u);
    if (xx < -103.278929903431851103f) return 0.0f;
    float x = xx;
    float z = __builtin_floorf(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    int n = (int)z;
    z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x +
          1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return scalePow2(z, n);
}
PT_DEV float pow_(float x, float y)
{
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : __uint_as_float(0x7fc00000u);
    if (x < 0.0f || x != x) return __uint_as_float(0x7fc00000u);
    return exp_(y * log_(x));
}

}  // namespace ptcore
