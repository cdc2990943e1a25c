// HIP kernels of the path-tracing core for gfx950 (wave64).
//
// Execution model: a persistent per-wave state machine.
//   * One lane owns one pixel for the whole launch and walks that pixel's FrameCount values
//     (samples) in order.  A wave covers an 8x8 pixel tile, a 256-thread block a 16x16 tile.
//   * Every lane is in one of four modes: NEW (start the next sample), SHADE (a closest-hit
//     query finished: shade it), TRAV (a BVH query is in flight), FINISH (accumulate the
//     sample).  The wave alternates between two phases:
//       - traversal phase: all TRAV lanes advance their BVH query one step per iteration (one
//         two-child node, or one triangle of a leaf).  Closest-hit and occlusion (shadow)
//         queries run in the same loop.  The phase ends when the share of lanes still
//         traversing falls below a threshold;
//       - shading phase: the lanes whose query finished shade their hit, or accumulate their
//         sample and regenerate a primary ray, and re-enter the traversal phase with fresh rays
//         next to the lanes whose query is still under way (their stack and cursor persist).
//     Lanes therefore never idle while a neighbour finishes a long traversal or a long path; a
//     wave leaves when every lane has finished every sample -- a condition every lane reaches.
//   * One shading pass emits both rays of a bounce.  The shader traces the shadow ray in the
//     middle of its bounce; everything it computes after an unoccluded shadow ray (light
//     radiance, BSDF value, MIS weight) does not depend on the ray's outcome, so it is
//     evaluated up front into a pending contribution that the shadow query merely gates.  The
//     shadow query and the extension query then run back to back inside the traversal phase
//     and no surface state has to survive a traversal.
//   * The per-lane stack of postponed far children lives in LDS (entry k of thread t at word
//     k*256 + t: consecutive lanes hit consecutive banks); the near child is followed directly.
//
// Same results as the reference shader's traversal (intersection.glsl:115-199,230-315):
// children are visited near-first with the shader's tie rule, leaves are scanned in order
// with a strict `<`, so the same triangle wins.  Two things differ without changing the
// result: sub-trees whose box entry distance lies beyond the best hit so far are skipped
// (the shader never prunes), and the stack never overflows (the shader's 16-entry stack
// aborts traversal when it does).
#pragma once
#include "shading.h"

namespace ptcore {

constexpr int kBlockThreads = 256;
// Entry distance must exceed the best hit by this factor before a sub-tree is skipped: box and
// triangle distances are rounded independently, the margin keeps a triangle whose computed
// distance ties or narrowly beats the current best from being culled with its box.
constexpr float kPruneMargin = 1.0009765625f;  // 1 + 2^-10

// Slab test with the shader's return convention (intersection.glsl:63-77): entry distance, or
// exit distance when the origin is inside, or -1 on a miss.  t0 is also returned for pruning.
PT_DEV float hitAabb(v3 o, v3 inv, float minx, float miny, float minz, float maxx, float maxy, float maxz, float& t0out)
{
    float fx = (maxx - o.x) * inv.x, fy = (maxy - o.y) * inv.y, fz = (maxz - o.z) * inv.z;
    float nx = (minx - o.x) * inv.x, ny = (miny - o.y) * inv.y, nz = (minz - o.z) * inv.z;
    float t1 = fmin_(fmax_(fx, nx), fmin_(fmax_(fy, ny), fmax_(fz, nz)));
    float t0 = fmax_(fmin_(fx, nx), fmax_(fmin_(fy, ny), fmin_(fz, nz)));
    t0out = t0;
    return (t1 >= t0) ? ((t0 > 0.0f) ? t0 : t1) : -1.0f;
}

// Primary ray: pathtracing.comp:150-200 (inverse(ViewMatrix) arrives precomputed in P.invView).
PT_DEV void mulMat4(const float* M, float x, float y, float z, float w, float out[4])
{
    for (int i = 0; i < 4; i++) out[i] = ((M[0 + i] * x + M[4 + i] * y) + M[8 + i] * z) + M[12 + i] * w;
}
PT_DEV void primaryRay(const DevParams& P, float u, float v, unsigned& rng, v3& ro, v3& rd)
{
    float vp[4];
    mulMat4(P.invProj, u * 2.0f - 1.0f, v * 2.0f - 1.0f, 0.0f, 1.0f, vp);
    v3 origin = mk3(vp[0], vp[1], vp[2]) / vp[3];
    v3 direction = normalize3(origin);
    float o4[4], d4[4];
    mulMat4(P.invView, origin.x, origin.y, origin.z, 1.0f, o4);
    mulMat4(P.invView, direction.x, direction.y, direction.z, 0.0f, d4);
    ro = mk3(o4[0], o4[1], o4[2]);
    rd = mk3(d4[0], d4[1], d4[2]);
    if (P.apertureSize != 0.0f) {  // thin lens (:166-187)
        const float* VM = P.view;
        v3 cameraForward = -mk3(VM[8], VM[9], VM[10]);
        v3 cameraUp = mk3(VM[4], VM[5], VM[6]);
        v3 cameraRight = cross(cameraUp, cameraForward);
        v3 focalPoint = ro + rd * P.focalLength;
        float r1 = randomValue(rng);
        float r2 = randomValue(rng);
        v3 h = hemispherePointCos(r1, r2);
        float ax = h.x * P.apertureShapeX * P.apertureSize;
        float ay = h.y * P.apertureShapeY * P.apertureSize;
        ro = ro + cameraRight * ax + cameraUp * ay;
        rd = normalize3(focalPoint - ro);
    }
}

PT_DEV float4 accumulate(float4 prev, v3 x, float weight, unsigned mode)
{
    if (mode == 0u)  // mix(prev, vec4(x, 1), weight): pathtracing.comp:248-260
        return make_float4(mix_(prev.x, x.x, weight), mix_(prev.y, x.y, weight), mix_(prev.z, x.z, weight),
                           mix_(prev.w, 1.0f, weight));
    return make_float4(prev.x + x.x, prev.y + x.y, prev.z + x.z, prev.w + 1.0f);
}

enum LaneMode : int { kModeNew = 0, kModeShade = 1, kModeTrav = 2, kModeFinish = 3, kModeDone = 4 };

template <int STACK, bool COUNT>
__global__ __launch_bounds__(kBlockThreads, 4) void renderKernel(DevParams P, DevScene S, float4* __restrict__ radianceImg,
                                                             float4* __restrict__ albedoImg,
                                                             float4* __restrict__ normalImg, DevCounters* counters)
{
    __shared__ int s_stack[STACK * kBlockThreads];
    const unsigned tid = threadIdx.x;
    int* stack = s_stack + tid;

    // Consecutive blocks go to different XCDs, so taking tiles in row-major order spreads every
    // image region -- cheap sky and expensive geometry alike -- evenly over the 8 XCDs.
    const unsigned tile = blockIdx.x;
    const unsigned tileX = tile % P.tilesX, tileY = tile / P.tilesX;
    const unsigned wave = tid >> 6, lane = tid & 63u;
    const unsigned px = tileX * 16u + (wave & 1u) * 8u + (lane & 7u);
    const unsigned py = tileY * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool inside = px < P.width && py < P.height;
    const unsigned pixelIndex = py * (unsigned)P.frameW + px;  // uint(FrameDimensions.x), pathtracing.comp:213
    const size_t texel = (size_t)py * P.width + px;

    unsigned nRays = 0, nShadow = 0, nHits = 0, nSamples = 0;
    unsigned nNodeVisits = 0, nTriTests = 0, maxStack = 0;

    // ---- path state
    unsigned k = 0;  // index of the next FrameCount value of this pixel
    const unsigned kEnd = inside ? P.frameCount : 0u;
    int mode = kModeNew;
    unsigned rng = 0, bounce = 0, frame = 0;
    v3 radiance = splat(0.0f), throughput = splat(1.0f), f = splat(1.0f);
    v3 albedoRes = splat(0.0f), normalRes = splat(0.0f);
    float pdfBsdf = 1.0f;
    // ---- ray / traversal state
    v3 ro = splat(0.0f), rd = splat(0.0f), inv = splat(0.0f);
    v3 extDir = splat(0.0f);  // direction of the extension ray waiting behind a shadow ray
    v3 pending = splat(0.0f);  // NEE contribution, added if the shadow ray is unoccluded
    bool isShadow = false, hasExt = false;
    int cur = 0, sp = 0, top = 0;  // `top` caches the newest stack entry; LDS holds the older ones
    unsigned leafPos = 0, leafEnd = 0;
    float bestT = kFltMax, bestU = 0.0f, bestV = 0.0f;
    int bestPrim = -1;

    // Starts a BVH query for the ray (ro, rd).
    auto beginQuery = [&]() {
        inv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        bestT = kFltMax;
        bestPrim = -1;
        sp = 0;
        cur = P.rootRef;
        leafPos = leafEnd = 0;
        if (cur < 0) {
            const unsigned bits = ~(unsigned)cur;
            leafPos = bits >> 3;
            leafEnd = leafPos + (bits & 7u) + 1u;
        }
        mode = kModeTrav;
    };

    for (;;) {
        // ================================================================ shading phase
        if (mode == kModeShade) {
            // one iteration of PathTrace's bounce loop (pathtracing.comp:25-140) around the
            // closest-hit result (bestPrim, bestT, bestU, bestV) of the ray (ro, rd)
            bool hasShadow = false;
            hasExt = false;
            mode = kModeFinish;
            if (bestPrim < 0) {
                float pdfLight = 0.0f;
                v3 colorLight = evaluateHdri(S, P.envW, P.envH, P.hdriW, P.hdriH, rd, pdfLight);
                float misWeight = 1.0f;
                if (bounce > 0u) {
                    float t = pdfBsdf * pdfBsdf;
                    misWeight = t / (pdfLight * pdfLight + t);
                }
                if (misWeight > 0.0f) radiance = radiance + misWeight * throughput * colorLight * f / pdfBsdf;
            } else {
                nHits++;
                // hit attributes of the winning triangle (intersection.glsl:41-57)
                const float w = 1.0f - bestU - bestV;
                const v3 P0 = ro + rd * bestT;
                const float4* shp = reinterpret_cast<const float4*>(S.shading + bestPrim);
                const float4 s0 = shp[0], s1 = shp[1], s2 = shp[2], s3 = shp[3], s4 = shp[4];
                const float uvx = s0.w * w + s2.w * bestU + s3.y * bestV;
                const float uvy = s1.w * w + s3.x * bestU + s3.z * bestV;
                v3 N = normalize3(mk3(s0.x, s0.y, s0.z) * w + mk3(s1.x, s1.y, s1.z) * bestU + mk3(s2.x, s2.y, s2.z) * bestV);
                const v3 Ng = mk3(s4.x, s4.y, s4.z);
                const unsigned meshIndex = __float_as_uint(s3.w);
                v3 tangent = splat(0.0f);
                if (S.materials[meshIndex].tex[kTexNormal] >= 0) {
                    const float4* tp = reinterpret_cast<const float4*>(S.triangles + bestPrim);
                    const float4 q1 = tp[1], q2 = tp[2];
                    const float dBx = s2.w - s0.w, dBy = s3.x - s1.w, dCx = s3.y - s0.w, dCy = s3.z - s1.w;
                    const float invTd = 1.0f / (dBx * dCy - dBy * dCx);
                    tangent = (mk3(q1.x, q1.y, q1.z) * dCy - mk3(q2.x, q2.y, q2.z) * dBy) * invTd;
                }
                const MaterialEval me = evaluateMaterial(S, P, meshIndex, uvx, uvy, N, tangent);
                const v3 V = -rd;
                const Surface surf = makeSurface(me, N, V);
                radiance = radiance + throughput * surf.emission;
                const v3 origin = offsetRay(P0, Ng);  // shared by the shadow and the extension ray (:66,:131)

                // next-event estimation towards the HDRI (pathtracing.comp:64-102), everything
                // but the occlusion test itself
                const v3 Ll = sampleHdri(S, P.envW, P.envH, rng);
                if (dot(N, Ll) > 0.0f) {
                    hasShadow = true;
                    pending = splat(0.0f);
                    float pdfLight = 0.0f;
                    v3 colorLight = evaluateHdri(S, P.envW, P.envH, P.hdriW, P.hdriH, Ll, pdfLight);
                    const Dir dl = makeDir(surf, Ll);
                    float pdfNee = 0.0f;
                    const v3 fNee = evaluateBsdf(surf, dl, pdfNee);
                    if (pdfNee > 0.0f) {
                        float t = pdfLight * pdfLight;
                        float misWeight = t / (pdfNee * pdfNee + t);
                        if (misWeight > 0.0f) pending = misWeight * throughput * colorLight * fNee / pdfLight;
                    }
                }

                // BSDF sampling, Russian roulette, next ray (pathtracing.comp:104-139)
                const v3 L = sampleBsdf(surf, rng);
                const Dir ds = makeDir(surf, L);
                pdfBsdf = 0.0f;
                f = evaluateBsdf(surf, ds, pdfBsdf);
                if (!(pdfBsdf <= 0.0f)) {
                    const float p = fmax_(throughput.x, fmax_(throughput.y, throughput.z));
                    if (!(randomValue(rng) >= p)) {
                        throughput = throughput * (1.0f / p);
                        throughput = throughput * (f / pdfBsdf);
                        if (bounce == 0u) {
                            albedoRes = surf.albedo;
                            normalRes = N;
                        }
                        bounce++;
                        hasExt = bounce < P.maxBounces;
                    }
                }
                // a shadow query whose pending contribution is exactly zero cannot change the
                // result and is not issued (a NaN contribution still is)
                hasShadow = hasShadow && !(pending.x == 0.0f && pending.y == 0.0f && pending.z == 0.0f);
                if (hasShadow) {
                    nShadow++;
                    isShadow = true;
                    extDir = L;
                    ro = origin;
                    rd = Ll;
                    beginQuery();
                } else if (hasExt) {
                    nRays++;
                    isShadow = false;
                    hasExt = false;
                    ro = origin;
                    rd = L;
                    beginQuery();
                }
            }
        }
        if (mode == kModeFinish) {
            const float weight = 1.0f / (float)frame;
            radianceImg[texel] = accumulate(radianceImg[texel], radiance, weight, P.accumulateMode);
            albedoImg[texel] = accumulate(albedoImg[texel], albedoRes, weight, P.accumulateMode);
            normalImg[texel] = accumulate(normalImg[texel], normalRes, weight, P.accumulateMode);
            k++;
            mode = kModeNew;
        }
        if (mode == kModeNew) {
            if (k < kEnd) {
                // main(): pathtracing.comp:206-245
                frame = P.frameFirst + k * P.frameStride;
                rng = pixelIndex + frame * 719393u;
                float u = ((float)px + 0.5f) / P.frameW, v = ((float)py + 0.5f) / P.frameH;
                if (P.antiAliasing != 0u) {
                    float r1 = randomValue(rng);
                    float r2 = randomValue(rng);
                    u = u + (r1 - 0.5f) / P.frameW;
                    v = v + (r2 - 0.5f) / P.frameH;
                }
                primaryRay(P, u, v, rng, ro, rd);
                radiance = splat(0.0f);
                throughput = splat(1.0f);
                f = splat(1.0f);
                pdfBsdf = 1.0f;
                albedoRes = splat(0.0f);
                normalRes = splat(0.0f);
                bounce = 0;
                isShadow = false;
                hasExt = false;
                nSamples++;
                nRays++;
                beginQuery();
            } else {
                mode = kModeDone;
            }
        }
        const unsigned long long travMask = __ballot(mode == kModeTrav);
        if (travMask == 0ull) break;
        // Leave the traversal phase once fewer than `threshold` lanes are traversing; lanes that
        // are done for good do not count against the wave.
        const int alive = __popcll(__ballot(mode != kModeDone));
        const int threshold = (alive * (int)P.travExitNum + 63) >> 6;

        // ============================================================== traversal phase
        for (;;) {
            if (mode == kModeTrav) {
                bool finished = false;
                // One 64-byte fetch per step, whichever record the lane needs: the current leaf's
                // next triangle (48 B + 16 B of the following record) or the current interior node.
                // Issuing it before the step diverges puts one memory round trip, not two, on the
                // wave's critical path.
                const bool atTriangle = leafPos < leafEnd;
                const float4* rec = atTriangle ? reinterpret_cast<const float4*>(S.triangles + leafPos)
                                               : reinterpret_cast<const float4*>(S.nodes + cur);
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                if (atTriangle) {
                    // ---- one triangle of the current leaf: Moller-Trumbore with the shader's
                    // expressions (intersection.glsl:26-40); edges and normal come precomputed
                    if (COUNT) nTriTests++;
                    const v3 n = mk3(r0.w, r1.w, r2.w);
                    const v3 ab = mk3(r1.x, r1.y, r1.z), ac = mk3(r2.x, r2.y, r2.z);
                    const v3 ao = ro - mk3(r0.x, r0.y, r0.z);
                    const v3 dao = cross(ao, rd);
                    const float det = -dot(rd, n);
                    const float invDet = 1.0f / det;
                    const float dst = dot(ao, n) * invDet;
                    const float u = dot(ac, dao) * invDet;
                    const float v = -dot(ab, dao) * invDet;
                    const float w = 1.0f - u - v;
                    const bool ok = det >= 1e-10f && dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
                    if (ok && (isShadow || dst < bestT)) {
                        bestT = dst;
                        bestU = u;
                        bestV = v;
                        bestPrim = (int)leafPos;
                    }
                    leafPos++;
                    if (isShadow && ok) {
                        finished = true;  // occluded
                    } else if (leafPos == leafEnd) {
                        if (sp == 0) {
                            finished = true;
                        } else {  // pop: the top entry is kept in a register
                            cur = top;
                            sp--;
                            if (sp > 0) top = stack[(sp - 1) * kBlockThreads];
                        }
                    }
                } else {
                    // ---- one interior node: both children's boxes from one 64-byte record
                    if (COUNT) nNodeVisits++;
                    float t0l, t0r;
                    const float dl = hitAabb(ro, inv, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, t0l);
                    const float dr = hitAabb(ro, inv, r2.x, r2.y, r2.z, r3.x, r3.y, r3.z, t0r);
                    const float limit = bestT * kPruneMargin;  // bestT stays FLT_MAX for shadow rays
                    const bool hl = dl > 0.0f && !(t0l > limit);
                    const bool hr = dr > 0.0f && !(t0r > limit);
                    const int left = __float_as_int(r0.w), right = __float_as_int(r1.w);
                    if (hl && hr) {
                        const bool leftFirst = dl < dr;  // intersection.glsl:169-180
                        if (sp > 0) stack[(sp - 1) * kBlockThreads] = top;
                        top = leftFirst ? right : left;
                        sp++;
                        if (COUNT) maxStack = (unsigned)sp > maxStack ? (unsigned)sp : maxStack;
                        cur = leftFirst ? left : right;
                    } else if (hl) {
                        cur = left;
                    } else if (hr) {
                        cur = right;
                    } else if (sp == 0) {
                        finished = true;
                    } else {
                        cur = top;
                        sp--;
                        if (sp > 0) top = stack[(sp - 1) * kBlockThreads];
                    }
                }
                if (!finished) {
                    if (cur < 0 && leafPos == leafEnd) {  // arrived at a leaf reference
                        const unsigned bits = ~(unsigned)cur;
                        leafPos = bits >> 3;
                        leafEnd = leafPos + (bits & 7u) + 1u;
                        cur = 0;
                    }
                } else if (isShadow) {
                    if (bestPrim < 0) radiance = radiance + pending;  // unoccluded
                    if (hasExt) {
                        nRays++;
                        isShadow = false;
                        hasExt = false;
                        rd = extDir;
                        beginQuery();
                    } else {
                        mode = kModeFinish;
                    }
                } else {
                    mode = kModeShade;
                }
            }
            if (__popcll(__ballot(mode == kModeTrav)) < threshold) break;
        }
    }

    // counters: wave-level reduction, one atomic per wave and counter
    auto waveSum = [](unsigned v) {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        return v;
    };
    const unsigned sRays = waveSum(nRays), sShadow = waveSum(nShadow), sHits = waveSum(nHits), sSamples = waveSum(nSamples);
    if (lane == 0) {
        atomicAdd(&counters->raysClosest, (unsigned long long)sRays);
        atomicAdd(&counters->raysAny, (unsigned long long)sShadow);
        atomicAdd(&counters->closestHits, (unsigned long long)sHits);
        atomicAdd(&counters->samples, (unsigned long long)sSamples);
    }
    if (COUNT) {
        const unsigned sNodes = waveSum(nNodeVisits), sTris = waveSum(nTriTests);
        unsigned m = maxStack;
        for (int off = 32; off > 0; off >>= 1) {
            unsigned o = __shfl_down(m, off, 64);
            m = o > m ? o : m;
        }
        if (lane == 0) {
            atomicAdd(&counters->nodeVisits, (unsigned long long)sNodes);
            atomicAdd(&counters->triangleTests, (unsigned long long)sTris);
            atomicMax(&counters->maxStack, m);
        }
    }
}

__global__ void scaleImagesKernel(float4* a, float4* b, float4* c, size_t n, float s)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 x = a[i], y = b[i], z = c[i];
        a[i] = make_float4(x.x * s, x.y * s, x.z * s, x.w * s);
        b[i] = make_float4(y.x * s, y.y * s, y.z * s, y.w * s);
        c[i] = make_float4(z.x * s, z.y * s, z.z * s, z.w * s);
    }
}

// Element-wise evaluation of the float32 building blocks, so tests can compare them bit for bit
// with a CPU evaluation of the same operation sequences.
__global__ void evalMathKernel(int fn, const float* a, const float* b, float* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    float r;
    switch (fn) {
    case 0: r = sin_(x); break;
    case 1: r = cos_(x); break;
    case 2: r = asin_(x); break;
    case 3: r = atan2_(x, y); break;
    case 4: r = pow_(x, y); break;
    case 5: r = fmin_(x, y); break;
    case 6: r = fmax_(x, y); break;
    case 7: r = x / y; break;
    case 8: r = sqrt_(x); break;
    default: r = rsqrt_(x); break;
    }
    out[i] = r;
}

}  // namespace ptcore
