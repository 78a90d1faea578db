// probe.hip -- TEST TOOL.  Evaluates the building blocks of srt_core.h on the GPU and on the host (the same
// header, compiled by the host pass of hipcc with the same flags) on random inputs and compares the results
// bit for bit.  It checks the arithmetic contract the parity tests rely on: IEEE division and square root, no
// contraction, identical operation order.  Prints one line per block: "<name> mismatches=<n> of <N>".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../simple_raytracer_amd/csrc/srt_core.h"

using namespace srt;

constexpr int N = 1 << 20;
constexpr int IN = 16, OUT = 8;

enum { P_DIV, P_SQRT, P_NORMALIZE, P_REFLECT, P_FRESNEL, P_POINT_LIGHT, P_SPHERE, P_TRI, P_RNG, P_HEMI, P_SKY_PACK, P_POW5, P_ATAN2, P_COUNT };
static const char* kNames[P_COUNT] = {"div", "sqrt", "normalize", "reflect", "fresnel", "point_light", "sphere", "tri", "rng",
                                      "hemisphere", "sky_pack", "pow5", "atan2"};

SRT_HD void eval(int which, const float* in, float* out, const float* unit)
{
    for (int i = 0; i < OUT; i++) out[i] = 0.0f;
    f3 a = mk3(in[0], in[1], in[2]), b = mk3(in[3], in[4], in[5]), c = mk3(in[6], in[7], in[8]);
    switch (which) {
        case P_DIV: out[0] = in[0] / in[1]; out[1] = 1.0f / in[2]; out[2] = in[3] / 100.0f; out[3] = (in[4] * 1e-30f) / (in[5] * 1e8f); break;
        case P_SQRT: out[0] = sqrt_rn(fabsf(in[0])); out[1] = sqrt_rn(fabsf(in[1]) * 1e-30f); out[2] = sqrt_rn(fabsf(in[2]) * 1e20f); break;
        case P_NORMALIZE: { f3 n = normalize3(a); out[0] = n.x; out[1] = n.y; out[2] = n.z; out[3] = dot3(a, b); f3 x = cross3(a, b); out[4] = x.x; out[5] = x.y; out[6] = x.z; break; }
        case P_REFLECT: { f3 r = reflect3(normalize3(a), add3(normalize3(b), scale3(c, 0.05f))); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        case P_FRESNEL: { f3 d = normalize3(a); f3 r = reflect3(d, add3(normalize3(b), scale3(c, 0.01f))); out[0] = fresnel_term(d, r); break; }
        case P_POINT_LIGHT: {
            Light L; L.position = scale3(a, 10.0f); L.brightness = 5.0f; L.color = mk3(0.0f, 0.0f, 0.0f);
            Material m; m.albedo = mk3(0.78f, 0.79f, 0.33f); m.metalness = in[12] > 0 ? 0.0f : 1.0f; m.smoothness = in[13] > 0 ? 0.995f : 0.45f; m.emission = mk3(0, 0, 0);
            f3 Nn = normalize3(b); f3 dir = normalize3(c);
            f3 P = mk3(in[9], in[10], in[11]);
            f3 r = point_light(L, P, Nn, dir, m, mk3(0.1f, 0.2f, 0.3f), in[14] > 0, fabsf(in[15]) * 5.0f, add3(P, scale3(Nn, 0.5f)), 0.3f);
            out[0] = r.x; out[1] = r.y; out[2] = r.z; break;
        }
        case P_SPHERE: {
            q4 s; s.x = 2.05f; s.y = 0.9f; s.z = -7.25f; s.w = 3.5f;
            f3 d = normalize3(mk3(in[0] * 0.6f, in[1] * 0.4f, -1.0f));
            float t = -1.0f; bool h = test_sphere(s, mk3(0, 0, 0), d, lensq3(d), 0.01f, INFINITY, t);
            out[0] = h ? 1.0f : 0.0f; out[1] = t;
            if (h) { f3 P = ray_at(mk3(0, 0, 0), d, t); float f = 1.0f / s.w; f3 n = normalize3(mk3((P.x - s.x) * f, (P.y - s.y) * f, (P.z - s.z) * f)); out[2] = n.x; out[3] = n.y; out[4] = n.z; }
            break;
        }
        case P_TRI: {
            q4 p0, p1; p0.x = in[0]; p0.y = in[1]; p0.z = in[2] - 3.0f; p0.w = in[3]; p1.x = in[4]; p1.y = in[5]; p1.z = in[6]; p1.w = in[7];
            f3 d = normalize3(mk3(in[9] * 0.3f, in[10] * 0.3f, -1.0f));
            float t = -1, u = -1, v = -1; bool h = test_tri(p0, p1, in[8], false, mk3(0, 0, 0), d, 0.01f, INFINITY, t, u, v);
            out[0] = h ? 1.0f : 0.0f; out[1] = t; out[2] = u; out[3] = v; break;
        }
        case P_RNG: {
            RngKey k; k.seed = fbits(in[0]); k.pixel = fbits(in[1]); k.sample = fbits(in[2]); k.node = fbits(in[3]);
            uint32_t w[4];
            rng_call(k, fbits(in[4]) & 0xFFFFu, w);
            f3 r = vector_from_word(w[0], unit); out[0] = r.x; out[1] = r.y; out[2] = r.z;
            f3 q = ranged_vector_from_word(w[1], 0.95f, unit); out[3] = q.x; out[4] = q.y; out[5] = q.z;
            out[6] = bitsf(w[2]); out[7] = bitsf(light_word(k, w, 5u)); break;
        }
        case P_HEMI: {
            RngKey k; k.seed = 1u; k.pixel = fbits(in[1]); k.sample = fbits(in[2]) & 1023u; k.node = 1u + (fbits(in[3]) & 255u);
            f3 r = random_hemisphere_vector(k, normalize3(a), unit); out[0] = r.x; out[1] = r.y; out[2] = r.z; break;
        }
        case P_SKY_PACK: { f3 s = sky_color(in[0]); out[0] = s.x; out[1] = s.y; out[2] = s.z; out[3] = bitsf(pack_color(in[1], in[2] * 2.0f, sqrt_rn(fabsf(in[3])))); break; }
        case P_POW5: { double x = (double)(1.0f - fabsf(in[0])); double x2 = x * x; out[0] = (float)(x2 * x2 * x); break; }
        case P_ATAN2: { out[0] = (float)(3.14159265358979323846 - atan2((double)fabsf(in[0]), (double)in[1])); break; }
    }
}

__global__ void probe_kernel(int which, const float* in, float* out, int n)
{
    __shared__ float unit[UNIT_TABLE];
    if (threadIdx.x < (unsigned)UNIT_TABLE) unit[threadIdx.x] = (float)((int)threadIdx.x - 100) / 100.0f;
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) eval(which, in + (size_t)i * IN, out + (size_t)i * OUT, unit);
}

int main()
{
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> uni(-1.0f, 1.0f);
    std::vector<float> in((size_t)N * IN), dev((size_t)N * OUT), host((size_t)N * OUT);
    for (auto& v : in) v = uni(rng);
    float *d_in, *d_out;
    if (hipMalloc(&d_in, in.size() * 4) != hipSuccess || hipMalloc(&d_out, dev.size() * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int which = 0; which < P_COUNT; which++) {
        hipLaunchKernelGGL(probe_kernel, dim3((N + 255) / 256), dim3(256), 0, 0, which, d_in, d_out, N);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
        hipMemcpy(dev.data(), d_out, dev.size() * 4, hipMemcpyDeviceToHost);
        float unit[UNIT_TABLE];
        fill_unit_table(unit);
        for (int i = 0; i < N; i++) eval(which, &in[(size_t)i * IN], &host[(size_t)i * OUT], unit);
        int bad = 0, first = -1;
        for (int i = 0; i < N; i++) {
            bool same = true;
            for (int k = 0; k < OUT; k++) {
                float a = dev[(size_t)i * OUT + k], b = host[(size_t)i * OUT + k];
                if (std::memcmp(&a, &b, 4) != 0 && !(a != a && b != b)) same = false;
            }
            if (!same) { bad++; if (first < 0) first = i; }
        }
        printf("%s mismatches=%d of %d", kNames[which], bad, N);
        if (first >= 0) {
            printf("  first at %d: dev=", first);
            for (int k = 0; k < OUT; k++) printf("%.9g ", dev[(size_t)first * OUT + k]);
            printf(" host=");
            for (int k = 0; k < OUT; k++) printf("%.9g ", host[(size_t)first * OUT + k]);
        }
        printf("\n");
        if (which != P_POW5 && which != P_ATAN2 && which != P_FRESNEL && which != P_POINT_LIGHT) bad_total += bad;
    }
    printf("contract_violations=%d\n", bad_total);
    return 0;
}
