/* ORACLE -- test infrastructure only.  See oracle_core.h for what this restates and what
 * is (un)pinned.  Build: oracle/Makefile -> oracle/_build/liboracle.so */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include "oracle.h"

#define REAL double
#define ORC_IS_F32 0
#define FN(x) x##_f64
#include "oracle_core.h"
#undef REAL
#undef ORC_IS_F32
#undef FN

#define REAL float
#define ORC_IS_F32 1
#define FN(x) x##_f32
#include "oracle_core.h"
#undef REAL
#undef ORC_IS_F32
#undef FN

int orc_sizeof_params(void) { return (int)sizeof(orc_params); }

/* ------------------------------------------------------------------------------------------
 * Reset sampler twin (float spec only).  The reference draws initial conditions from NumPy's
 * global MT19937 (misc.py:49-63 -> rps generate_initial_conditions, Appendix A.7;
 * MaterialTransport.py:99-100); the device cannot share that stream, so sim_spec_v0 gives every
 * (global env index, episode) its own Philox4x32-10 stream and this file restates the sampler
 * the HIP kernel runs: partial Fisher-Yates over the grid cells, heading uniform in [-pi, pi),
 * zone loads int(mean + std * z) with z from Box-Muller on the spec'd log / sincos.
 * Distribution parity with the reference is checked statistically (tests/test_reset.py). */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct { uint32_t k0, k1, c0, c1, c2, next; } draws_t;

static uint32_t draw_u32(draws_t *d) {
    uint32_t blk[4];
    philox4x32_10(d->c0, d->c1, d->c2, d->next >> 2, d->k0, d->k1, blk);
    return blk[(d->next++) & 3u];
}

static float log_spec_f32(float x) {
    int e;
    float m = frexpf(x, &e);
    if (m < 0.707106781186547524f) { e = e - 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = __builtin_fmaf(m, 7.0376836292e-2f, -1.1514610310e-1f);
    y = __builtin_fmaf(m, y, 1.1676998740e-1f);
    y = __builtin_fmaf(m, y, -1.2420140846e-1f);
    y = __builtin_fmaf(m, y, 1.4249322787e-1f);
    y = __builtin_fmaf(m, y, -1.6668057665e-1f);
    y = __builtin_fmaf(m, y, 2.0000714765e-1f);
    y = __builtin_fmaf(m, y, -2.4999993993e-1f);
    y = __builtin_fmaf(m, y, 3.3333331174e-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(z, -0.5f, y);
    float r = m + y;
    return __builtin_fmaf(fe, 0.693359375f, r);
}

static int normal_int(draws_t *d, float mean, float stdv) {
    uint32_t r1 = draw_u32(d), r2 = draw_u32(d);
    float u1 = (float)((r1 >> 8) + 1u) * 5.9604644775390625e-08f;
    float u2 = (float)(r2 >> 8) * 5.9604644775390625e-08f;
    float rad = __builtin_sqrtf(-2.0f * log_spec_f32(u1));
    float sn, cs;
    sincos_f32(u2 * 6.283185482025146484375f - 3.1415927410125732421875f, &sn, &cs);
    return (int)(mean + stdv * (rad * cs));
}

static void sample_cells(draws_t *d, const orc_grid *g, int count, float *outx, float *outy, int stride) {
    uint8_t perm[64];
    int C = g->nx * g->ny;
    for (int i = 0; i < C; ++i) perm[i] = (uint8_t)i;
    for (int i = 0; i < count; ++i) {
        uint32_t r = draw_u32(d);
        int j = i + (int)(((uint64_t)r * (uint32_t)(C - i)) >> 32);
        uint8_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        int cell = perm[i] + 1 /* rps: choice(...) + 1 before divmod */, cx = cell / g->ny, cy = cell - cx * g->ny;
        float x = (float)cx * g->spacing - g->w2, y = (float)cy * g->spacing - g->h2;
        outx[i * stride] = (x + g->ox1) + g->ox2;
        outy[i * stride] = (y + g->oy1) + g->oy2;
    }
}

/* One env: writes poses [3][N], prey_loc [P][2], zone_load [2].  episode = the env's reset_count. */
/* ArcticTransport (ArcticTransport.py:56-82): fixed poses; grid[96] uniform {0,1,2} from draws 0..95
 * (row-major), goal column 1..11 from draw 96, 2x2 goal block in rows 0-1, row 7 columns 1..10 cleared. */
void orc_reset_arctic_f32(uint64_t seed, uint64_t global_env, uint32_t episode, float *poses, uint8_t *grid,
                          int32_t *goal_col) {
    draws_t d = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)global_env, (uint32_t)(global_env >> 32), episode, 0};
    uint32_t cell[96];
    for (int i = 0; i < 96; ++i) cell[i] = draw_u32(&d);
    int gc = 1 + (int)(((uint64_t)draw_u32(&d) * 11u) >> 32);
    for (int i = 0; i < 96; ++i) {
        int row = i / 12, col = i % 12;
        int val = (int)(((uint64_t)cell[i] * 3u) >> 32);
        if (row <= 1 && (col == gc || col == gc - 1)) val = 3;
        if (row == 7 && col >= 1 && col <= 10) val = 0;
        grid[i] = (uint8_t)val;
    }
    static const float sx[4] = {-0.3f, 0.3f, -0.9f, 0.9f};
    for (int a = 0; a < 4; ++a) {
        poses[a] = sx[a];
        poses[4 + a] = -0.8f;
        poses[8 + a] = 1.57079637050628662109375f;
    }
    *goal_col = gc;
}

void orc_reset_env_f32(const orc_reset_params *p, uint64_t seed, uint64_t global_env, uint32_t episode, float *poses,
                       float *prey_loc, int32_t *zone_load) {
    draws_t d = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)global_env, (uint32_t)(global_env >> 32), episode, 0};
    int N = p->n_agents;
    if (p->scenario == ORC_SCN_MT) {
        zone_load[0] = normal_int(&d, p->zone1_mean, p->zone1_std);
        zone_load[1] = normal_int(&d, p->zone2_mean, p->zone2_std);
    }
    sample_cells(&d, &p->agent_grid, N, poses, poses + N, 1);
    for (int i = 0; i < N; ++i) {
        float th = (float)(draw_u32(&d) >> 8) * 5.9604644775390625e-08f * 6.283185482025146484375f -
                   3.1415927410125732421875f;
        poses[2 * N + i] = p->keep_theta ? th : 0.0f;
    }
    if (p->scenario == ORC_SCN_PCP || p->scenario == ORC_SCN_SIMPLE)
        sample_cells(&d, &p->prey_grid, p->num_prey, prey_loc, prey_loc + 1, 2);
}
