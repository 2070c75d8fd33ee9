// tirt_api.hip -- the C-ABI of include/tirt.h: context, uploads/downloads, camera, film,
// tone map, Scene.process_normal / total_area kernels, known-answer-test kernels, stats.
#include "tirt_internal.h"
#include "tirt_spectral.h"
#include <stddef.h>
#include <mutex>
#include <string.h>

namespace tirt {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

SceneView scene_view(const tirt_ctx *c)
{
    SceneView s;
    s.vertex = c->vertex.as<float>(); s.primitive = c->primitive.as<int>(); s.material = c->material.as<float>();
    s.shape = c->shape.as<float>(); s.light = c->light.as<int>(); s.env = c->env.as<int>();
    s.mat_lrgb = c->mat_lrgb.as<float>(); s.shade_rec = c->shade_rec.as<float4>();
    s.n = c->n; s.light_count = c->light_count; s.env_w = c->env_w; s.env_h = c->env_h; s.env_power = c->env_power;
    return s;
}
BvhView bvh_view(const tirt_ctx *c)
{
    BvhView b;
    b.wnode = c->wnode.as<float4>(); b.tri = c->tri.as<float4>();
    b.cnode = c->cnode.as<uint4>(); b.top_count = c->wide_nodes + c->n_far_nodes < TR_TOP_SLOTS ? c->wide_nodes + c->n_far_nodes : TR_TOP_SLOTS; b.compact = c->compact.as<float>(); b.cparent = c->cparent.as<int>();
    for (int k = 0; k < 3; k++) { b.grid_min[k] = c->grid_min[k]; b.cell[k] = c->grid_cell[k]; b.inv_cell[k] = c->grid_inv_cell[k]; b.inv_extent[k] = c->grid_inv_extent[k]; }
    for (int k = 0; k < 3; k++) { b.root_min[k] = c->root_min[k]; b.root_max[k] = c->root_max[k]; }
    b.root_code = c->root_code;
    b.root_qcode = c->root_code;        // >= 0: wide node 0
    b.far_qcode = c->n_far_nodes ? c->wide_nodes : c->root_code;
    return b;
}
int flush_pending(tirt_ctx *c)
{
    if (!c->pend.valid) return 0;
    c->pend.valid = false;
    return pt_render(c, c->pend.begin, c->pend.count, c->pend.seed, c->pend.max_depth, c->pend.stack_size, c->pend.flags,
                     c->pend.spectral ? (const SpecView *)c->spec_view : nullptr);
}
int sync_all(tirt_ctx *c)
{
    if (int rc = flush_pending(c)) return rc;
    for (Lane &L : c->lanes) if (L.stream) TIRT_HIP(hipStreamSynchronize(L.stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    c->batches_since_sync = 0;
    return 0;
}
int ensure_counters(tirt_ctx *c)
{
    if (!c->dev_counters.p) {
        if (c->dev_counters.ensure(sizeof(DevCounters))) return TIRT_ERR_HIP;
        TIRT_HIP(hipMemsetAsync(c->dev_counters.p, 0, sizeof(DevCounters), c->stream));
    }
    return 0;
}

// ---- per-material srgb_to_lrgb(colour) table (integrator/PT_RGB.py:86 evaluates it per path vertex) ----
__global__ void k_material_lrgb(const float *material, int nm, float *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nm) return;
    const float *m = material + (size_t)i * MAT_VEC;
    v3 c = srgb_to_lrgb(V(m[2], m[3], m[4]));
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}
static int refresh_material_table(tirt_ctx *c)
{
    if (c->mat_lrgb.ensure(sizeof(float) * 3 * (size_t)c->nm)) return TIRT_ERR_HIP;
    hipLaunchKernelGGL(k_material_lrgb, dim3((c->nm + 63) / 64), dim3(64), 0, c->stream, c->material.as<float>(), c->nm, c->mat_lrgb.as<float>());
    return 0;
}

// ---- 128-byte shading record per primitive (tirt_device.h, hit_attributes_rec) -----------------------
__global__ void k_shade_records(SceneView s, float4 *rec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    const int *pr = s.primitive + (size_t)i * PRI_VEC;
    float4 *r = rec + (size_t)i * 8;
    const float mat = __int_as_float(pr[2]);
    if (pr[0] == PRIMITIVE_TRI) {
        const v3 v1 = vtx_pos(s, pr[1]), v2 = vtx_pos(s, pr[1] + 1), v3_ = vtx_pos(s, pr[1] + 2);
        const v3 n1 = vtx_nor(s, pr[1]), n2 = vtx_nor(s, pr[1] + 1), n3 = vtx_nor(s, pr[1] + 2);
        r[0] = make_float4(v1.x, v1.y, v1.z, mat); r[1] = make_float4(v2.x, v2.y, v2.z, __int_as_float(PRIMITIVE_TRI));
        r[2] = make_float4(v3_.x, v3_.y, v3_.z, 0.0f);
        r[3] = make_float4(n1.x, n1.y, n1.z, 0.0f); r[4] = make_float4(n2.x, n2.y, n2.z, 0.0f); r[5] = make_float4(n3.x, n3.y, n3.z, 0.0f);
    } else {
        const float *sh = s.shape + (size_t)pr[1] * SHA_VEC;
        r[0] = make_float4(sh[1], sh[2], sh[3], mat); r[1] = make_float4(sh[4], sh[0], 0.0f, __int_as_float(2));
        r[2] = r[3] = r[4] = r[5] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    r[6] = r[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
int ensure_shade_records(tirt_ctx *c)
{
    if (c->shade_rec_valid && c->shade_rec.p) return 0;
    if (c->shade_rec.ensure(sizeof(float4) * 8 * (size_t)c->n)) return TIRT_ERR_HIP;
    hipLaunchKernelGGL(k_shade_records, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, scene_view(c), c->shade_rec.as<float4>());
    c->shade_rec_valid = true;
    return 0;
}

// ---- Scene.total_area (Scene.py:747-750): serial so the f32 sum order is defined --------------
__global__ void k_total_area(SceneView s, int count, float *out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float a = 0.0f;
        for (int i = 0; i < count; i++) a += get_prim_area(s, s.light[i]);
        *out = a;
    }
}

// ---- Scene.process_normal (Scene.py:754-798): BVH point query per vertex ------------------------
__global__ void k_smooth_normal(SceneView s, int nv, const float *compact, const int *vertex_index, float *smooth)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    constexpr int MAX_STACK_SIZE = 32;              // Scene.py:19
    int stack[MAX_STACK_SIZE + 2];
    v3 v = vtx_pos(s, i);
    v3 n = normalized(vtx_nor(s, i));
    int f = vertex_index[i];
    v3 sm = (n * get_prim_angle(s, f, v)) * get_prim_area(s, f);
    stack[0] = 0;
    int stack_pos = 0;
    while ((stack_pos >= 0) & (stack_pos < MAX_STACK_SIZE)) {
        int node = stack[stack_pos];
        stack_pos -= 1;
        const float *cn = compact + (size_t)node * CPN_VEC;
        if ((((int)cn[0]) & 1) == 1) {
            int prim = (int)cn[1];
            const int *pr = s.primitive + (size_t)prim * PRI_VEC;
            if (pr[0] == PRIMITIVE_TRI) {
                for (int j = 0; j < 3; j++) {
                    int nb = j + pr[1];
                    if (i != nb) {
                        v3 nvp = vtx_pos(s, nb);
                        v3 nn = normalized(vtx_nor(s, nb));
                        if ((int)(norm(v - nvp) < 0.000001f) & (int)(dot(nn, n) > 0.5f)) {      // (Scene.py:781: both sides evaluated, as the reference's `&` does)
                            float angle = get_prim_angle(s, prim, nvp);
                            sm = sm + (nn * angle) * get_prim_area(s, prim);
                        }
                    }
                }
            }
        } else {
            if ((v.x >= cn[2]) & (v.y >= cn[3]) & (v.z >= cn[4]) & (v.x <= cn[5]) & (v.y <= cn[6]) & (v.z <= cn[7])) {
                stack_pos += 1; stack[stack_pos] = node + 1;
                stack_pos += 1; stack[stack_pos] = (int)cn[1];
            }
        }
    }
    smooth[3 * i] = sm.x; smooth[3 * i + 1] = sm.y; smooth[3 * i + 2] = sm.z;
}
__global__ void k_write_normal(float *vertex, int nv, const float *smooth)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    v3 nn = normalized(V(smooth[3 * i], smooth[3 * i + 1], smooth[3 * i + 2]));
    float *p = vertex + (size_t)i * VER_VEC;
    p[3] = nn.x; p[4] = nn.y; p[5] = nn.z;
}

// UtilsFunc.py:583-586
__global__ void k_tone_map(const float *hdr, float *rgb, long nvals, float exposure)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvals) return;
    rgb[i] = lrgb_to_srgb1(tone_aces1(hdr[i] * exposure));
}

// ---- known-answer-test kernels ------------------------------------------------------------------------
__global__ void k_kat_math(int fn, const float *x, const float *y, float *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (fn) {
        case 0: r = tm_sin(x[i]); break;
        case 1: r = tm_cos(x[i]); break;
        case 2: r = tm_exp(x[i]); break;
        case 3: r = tm_log(x[i]); break;
        case 4: r = tm_pow(x[i], y[i]); break;
        case 5: r = tm_atan2(x[i], y[i]); break;
        case 6: r = tm_acos(x[i]); break;
        case 7: r = tm_sqrt(x[i]); break;
        case 8: r = x[i] / y[i]; break;
        case 9: r = tm_rand(tm_f2u(x[i]), tm_f2u(y[i]), 3u, 5u); break;
        case 10: { float sn, cs; tm_sincos(x[i], &sn, &cs); r = sn; } break;
        case 11: { float sn, cs; tm_sincos(x[i], &sn, &cs); r = cs; } break;
    }
    out[i] = r;
}
__global__ void k_kat_brdf(int which, const float *in, int in_stride, float *out, int out_stride, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = in + (size_t)i * in_stride;
    float *o = out + (size_t)i * out_stride;
    if (which == 0) {
        float pdf; float f = disney_evaluate_pdf(a, V(a[10], a[11], a[12]), V(a[13], a[14], a[15]), V(a[16], a[17], a[18]), pdf);
        o[0] = f; o[1] = pdf;
    } else if (which == 1) {
        v3 r = disney_sample(a, V(a[10], a[11], a[12]), V(a[13], a[14], a[15]), a[16], a[17], a[18]);
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else if (which == 2) {
        float fb; v3 r = glass_sample(a, V(a[10], a[11], a[12]), V(a[13], a[14], a[15]), a[16], fb);
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = fb;
    } else if (which == 3) {
        v3 r = offset_ray(V(a[0], a[1], a[2]), V(a[3], a[4], a[5]));
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else if (which == 4) {
        v3 r = cosine_sample_hemisphere(a[0], a[1]); o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else if (which == 5) {
        map_to_disk(a[0], a[1], o[0], o[1]);
    } else if (which == 6) {
        o[0] = power_heuristic(a[0], a[1]);
    } else if (which == 7) {
        v3 r = inverse_transform(V(a[0], a[1], a[2]), V(a[3], a[4], a[5])); o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else if (which == 8) {
        v3 r = srgb_to_lrgb(V(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else if (which == 9) {
        o[0] = lrgb_to_srgb1(a[0]); o[1] = lrgb_to_srgb1(a[1]); o[2] = lrgb_to_srgb1(a[2]);
    } else if (which == 10) {
        o[0] = tone_aces1(a[0]); o[1] = tone_aces1(a[1]); o[2] = tone_aces1(a[2]);
    } else if (which == 11) {
        float suc; v3 r = refract_(V(a[0], a[1], a[2]), V(a[3], a[4], a[5]), a[6], suc); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = suc;
    } else if (which == 12) {
        o[0] = schlick(a[0], a[1]);
    } else if (which == 13) {
        o[0] = gtr2(a[0], a[1]);
    } else if (which == 14) {
        o[0] = smithg_ggx(a[0], a[1]);
    } else if (which == 15) {
        o[0] = schlick_fresnel(a[0]);
    } else if (which == 16) {
        float fb; v3 r = glass_sample_lambda(V(a[0], a[1], a[2]), V(a[3], a[4], a[5]), a[6], a[7], fb); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = fb;
    } else if (which == 17) {
        CameraView cv; for (int k = 0; k < 12; k++) cv.view_inv[k] = a[k];
        cv.eye[0] = cv.eye[1] = cv.eye[2] = 0.0f; cv.fx = a[16]; cv.fy = a[17]; cv.cx = a[18]; cv.cy = a[19];
        v3 r = camera_ray_direction(cv, (int)a[20], (int)a[21], a[22], a[23]); o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } else if (which == 18) {
        const RayCtx r = make_ray(V(a[0], a[1], a[2]), V(a[3], a[4], a[5])); float tn;
        const int full = slabs(r, a[6], a[7], a[8], a[9], a[10], a[11], tn);
        o[0] = (float)full;
        o[1] = ray_has_parallel_axis(r) ? (float)full : (float)slabs_fast(r, a[6], a[7], a[8], a[9], a[10], a[11], tn);     // the branch-free form k_trace uses where it may
    }
}

// ---- bench helper: ceiling of scattered 64-byte record gathers (the node-fetch pattern of k_trace) ----
TD uint32_t mg_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void k_micro_gather(const float4 *rec, uint32_t nrec, int iters, float *out)
{
    uint32_t s = mg_mix(blockIdx.x * 256 + threadIdx.x + 1);
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        s = mg_mix(s + it);
        const uint32_t idx = (uint32_t)(((unsigned long long)s * nrec) >> 32);
        const float4 *p = rec + (size_t)idx * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st)
{
    if (b.ensure(bytes)) return TIRT_ERR_HIP;
    if (bytes) TIRT_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
    return 0;
}

static void drain_render_events(tirt_ctx *c)
{
    for (auto &pr : c->ev_pool) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->ms_render += ms;
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    c->ev_pool.clear();
}

}  // namespace tirt

using namespace tirt;

extern "C" {

const char *tirt_last_error(void) { return g_error.c_str(); }
int tirt_version(void) { return 100; }

int tirt_device_count(int *out)
{
    TIRT_REQUIRE(out, "tirt_device_count: null out");
    TIRT_HIP(hipGetDeviceCount(out));
    return TIRT_OK;
}

int tirt_create(int device_id, tirt_ctx **out)
{
    TIRT_REQUIRE(out, "tirt_create: null out");
    int ndev = 0;
    TIRT_HIP(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) { set_error("tirt_create: device " + std::to_string(device_id) + " of " + std::to_string(ndev)); return TIRT_ERR_ARG; }
    TIRT_HIP(hipSetDevice(device_id));
    tirt_ctx *c = new tirt_ctx();
    c->device = device_id;
    TIRT_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    TIRT_HIP(hipEventCreate(&c->ev0));
    TIRT_HIP(hipEventCreate(&c->ev1));
    TIRT_HIP(hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    for (Lane &L : c->lanes) {
        TIRT_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        TIRT_HIP(hipEventCreateWithFlags(&L.film_done, hipEventDisableTiming));
    }
    memset(&c->cam, 0, sizeof(c->cam));
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, device_id) == hipSuccess && optin > 0) c->lds_optin = (size_t)optin;
    else { (void)hipGetLastError(); if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && optin > 0) c->lds_optin = (size_t)optin; }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) {
        const int g = 5 * cus < 2048 ? 5 * cus : 2048;            // five 256-thread k_trace blocks per CU (tirt_internal.h, TR_TOP_CAP)
        c->tr_grid = g; c->tr_grid_alone = g; c->cu_count = cus;
    } else (void)hipGetLastError();
    *out = c;
    return TIRT_OK;
}

void tirt_destroy(tirt_ctx *c)
{
    if (!c) return;
    if (c->comm) { tirt_ctx *one[1] = {c}; (void)tirt_comm_destroy(one, 1); }
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    drain_render_events(c);
    DevBuf *bufs[] = {&c->vertex, &c->primitive, &c->material, &c->shape, &c->light, &c->env, &c->mat_lrgb, &c->shade_rec, &c->morton_unsorted, &c->keys_a,
                      &c->keys_b, &c->vals_a, &c->vals_b, &c->hist, &c->morton_sorted, &c->bvh_node, &c->compact, &c->parent,
                      &c->flag, &c->subtree, &c->build_status, &c->leaf_compact, &c->wnode, &c->tri, &c->prim_slot, &c->cnode, &c->cparent, &c->csize, &c->wide_queue, &c->wide_levels, &c->sah_compact, &c->sah_csize, &c->sah_parent, &c->wide_dp, &c->sah_box, &c->sah_idx, &c->sah_tasks, &c->sah_counts, &c->hdr, &c->rgb,
                      &c->counters_mem, &c->spill, &c->tr_rays,
                      &c->tr_out, &c->tr_prim, &c->tr_counts, &c->dev_counters, &c->bdpt_px, &c->timeline, &c->pvb_set[0].count, &c->pvb_set[0].cand, &c->pvb_set[0].bound, &c->pvb_set[1].count, &c->pvb_set[1].cand, &c->pvb_set[1].bound, &c->pvb_stat, &c->pvb_tmp};
    for (DevBuf *b : bufs) b->release();
    for (auto &bl : c->bd) {
        DevBuf *bb[] = {&bl.items, &bl.state, &bl.rays, &bl.hits, &bl.qidx, &bl.ctr, &bl.rad};
        for (DevBuf *b : bb) b->release();
        if (bl.delta_done) (void)hipEventDestroy(bl.delta_done);
        if (bl.film_done) (void)hipEventDestroy(bl.film_done);
    }
    for (Lane &L : c->lanes) {
        DevBuf *lb[] = {&L.path_mem, &L.counters_mem, &L.spill};
        for (DevBuf *b : lb) b->release();
        if (L.film_done) (void)hipEventDestroy(L.film_done);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    c->spec_mem.release(); c->spec_dev.release();
    if (c->spec_view) { delete (SpecView *)c->spec_view; c->spec_view = nullptr; }
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    (void)hipEventDestroy(c->ev0); (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

// film readers/writers on the main stream run after the last film update of the render lanes
#define AFTER_RENDER(c)                                                            \
    do { if ((c)->last_film) TIRT_HIP(hipStreamWaitEvent((c)->stream, (c)->last_film, 0)); } while (0)

#define CTX_NOFLUSH(c)                                                             \
    TIRT_REQUIRE(c, "null context");                                               \
    TIRT_HIP(hipSetDevice((c)->device))
// every entry point except tirt_pt_rgb_render first submits the render calls still pending
#define CTX(c)                                                                     \
    CTX_NOFLUSH(c);                                                                \
    do { if (int rc__ = flush_pending(c)) return rc__; } while (0)

int tirt_sync(tirt_ctx *c)
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

int tirt_set_option(tirt_ctx *c, const char *name, double value)
{
    CTX(c);
    TIRT_REQUIRE(name, "tirt_set_option: null name");
    if (!strcmp(name, "time_kernels")) { c->time_kernels = value != 0.0; return TIRT_OK; }
    if (!strcmp(name, "overlap_lanes")) { TIRT_REQUIRE(value >= 1.0 && value <= (double)TIRT_MAX_LANES, "overlap_lanes: 1..8"); if (sync_all(c)) return TIRT_ERR_HIP; c->n_lanes = (int)value; return TIRT_OK; }
#ifndef TIRT_EXPERIMENTS
    if (!strcmp(name, "wide_collapse")) {
        TIRT_REQUIRE(value == 0.0, "this option is an experiment (cost-optimal wide collapse): build with -DTIRT_EXPERIMENTS (make experiments)");
        return TIRT_OK;
    }
#endif
    if (!strcmp(name, "split_lone_batch")) { TIRT_REQUIRE(value >= 0.0 && value <= 8.0, "split_lone_batch: 0 (off) or the number of parts, 2..8"); c->split_lone = (int)value; return TIRT_OK; }
    if (!strcmp(name, "wide_collapse")) { TIRT_REQUIRE(value == 0.0 || value == 1.0, "wide_collapse: 0 (greedy) or 1 (cost-optimal)"); c->wide_dp_on = (int)value; return TIRT_OK; }
    if (!strcmp(name, "traversal_tree")) {       // takes effect at the next tirt_lbvh_build
        TIRT_REQUIRE(value == 0.0 || value == 1.0, "traversal_tree: 0 (the reference's LBVH) or 1 (binned SAH)");
        c->use_sah = (int)value; return TIRT_OK;
    }
    if (!strcmp(name, "plan_batches")) { TIRT_REQUIRE(value >= 0.0 && value <= 64.0, "plan_batches: 0 (automatic) .. 64"); if (flush_pending(c)) return TIRT_ERR_HIP; c->plan_nb = (int)value; return TIRT_OK; }
    if (!strcmp(name, "plan_lanes")) { TIRT_REQUIRE(value >= 0.0 && value <= TIRT_MAX_LANES, "plan_lanes: 0 (automatic) .. the lane count"); if (flush_pending(c)) return TIRT_ERR_HIP; c->plan_lanes = (int)value; return TIRT_OK; }
    if (!strcmp(name, "job_frames")) { TIRT_REQUIRE(value >= 0.0 && value <= 1.0e9, "job_frames out of range"); c->job_frames = (long)value; return TIRT_OK; }
    if (!strcmp(name, "merge_paths")) { TIRT_REQUIRE(value >= 0.0 && value <= 1.0e9, "merge_paths out of range"); c->merge_paths = (size_t)value; c->merge_user = true; return TIRT_OK; }
    if (!strcmp(name, "batch_paths")) {
        TIRT_REQUIRE(value >= 1.0 && value <= 1.0e9, "tirt_set_option: batch_paths out of range");
        c->batch_paths = (size_t)value; c->batch_user = true; return TIRT_OK;
    }
    if (!strcmp(name, "trace_lds_depth")) {
        TIRT_REQUIRE(value >= 12 && value <= 64, "trace_lds_depth: 12..64");
        TIRT_REQUIRE(trace_lds_bytes((int)value) <= c->lds_optin, "trace_lds_depth: stacks + tree top exceed the LDS a block can have on this device");
        c->tr_lds_depth = (int)value; return TIRT_OK;
    }
    if (!strcmp(name, "bdpt_stack_size")) { TIRT_REQUIRE(value >= 16 && value <= 4096, "bdpt_stack_size: 16..4096"); c->bdpt_stack = (int)value; return TIRT_OK; }
    if (!strcmp(name, "trace_timeline")) { c->timeline_arm = (int)value; c->timeline_waves = 0; return TIRT_OK; }
    if (!strcmp(name, "trace_refill_min")) { TIRT_REQUIRE(value >= 1 && value <= 64, "trace_refill_min: 1..64"); c->tr_refill_min = (int)value; return TIRT_OK; }
    if (!strcmp(name, "trace_node_min")) { TIRT_REQUIRE(value >= 1 && value <= 64, "trace_node_min: 1..64"); c->tr_node_min = (int)value; return TIRT_OK; }
    if (!strcmp(name, "bdpt_bounded")) { c->bdpt_bounded = value != 0.0 ? 1 : 0; return TIRT_OK; }
    if (!strcmp(name, "bdpt_state_fill")) { TIRT_REQUIRE(value == 0.0 || value == 1.0 || value == 2.0, "bdpt_state_fill: 0 (none), 1 (zeros) or 2 (poison)"); c->bdpt_state_fill = (int)value; return TIRT_OK; }
    if (!strcmp(name, "primary_beams")) { TIRT_REQUIRE(value == 0.0 || value == 1.0, "primary_beams: 0 or 1"); if (flush_pending(c)) return TIRT_ERR_HIP; c->primary_beams = (int)value; return TIRT_OK; }
    // "primary_beams_rebuild": forget the camera rays' candidate lists -- the next batch that uses lists makes them again (bench.py: a list build inside its clock)
    if (!strcmp(name, "primary_beams_rebuild")) { if (flush_pending(c)) return TIRT_ERR_HIP; c->pvb_valid = false; return TIRT_OK; }
    if (!strcmp(name, "primary_beams_diag")) { if (flush_pending(c)) return TIRT_ERR_HIP; c->pvb_diag = value != 0.0; return TIRT_OK; }
    if (!strcmp(name, "primary_beams_min_frames")) { TIRT_REQUIRE(value >= 1.0 && value <= 1.0e6, "primary_beams_min_frames out of range"); if (flush_pending(c)) return TIRT_ERR_HIP; c->primary_beams_min_frames = (int)value; return TIRT_OK; }
    if (!strcmp(name, "bdpt_lanes")) { TIRT_REQUIRE(value >= 1.0 && value <= 4.0, "bdpt_lanes: 1..4"); if (sync_all(c)) return TIRT_ERR_HIP; c->bdpt_lanes = (int)value; return TIRT_OK; }
    if (!strcmp(name, "bdpt_batch_items")) { TIRT_REQUIRE(value >= 1.0 && value <= 1.0e9, "bdpt_batch_items out of range"); c->bdpt_batch_items = (size_t)value; return TIRT_OK; }
    if (!strcmp(name, "bdpt_mem_budget")) { TIRT_REQUIRE(value >= 0, "bdpt_mem_budget: bytes >= 0"); c->bdpt_mem_budget = (size_t)value; return TIRT_OK; }
    if (!strcmp(name, "trace_slices")) {
        const int iv = (int)value;
        TIRT_REQUIRE(iv >= 1 && iv <= 64 && (iv & (iv - 1)) == 0, "trace_slices: power of two, 1..64");
        int l = 0; while ((1 << l) < iv) l++;
        c->tr_slice_log2 = l; return TIRT_OK;
    }
    if (!strcmp(name, "shade_grid")) { TIRT_REQUIRE(value >= 1 && value <= 65536, "shade_grid: 1..65536"); c->sh_grid = (int)value; c->grid_user = true; return TIRT_OK; }
    if (!strcmp(name, "path_order_blocks")) { c->path_order_blocks = value != 0.0; return TIRT_OK; }
    if (!strcmp(name, "slices_contiguous")) { c->slices_contiguous = value != 0.0; return TIRT_OK; }
    if (!strcmp(name, "trace_grid_alone")) { TIRT_REQUIRE(value >= 1 && value <= 16384, "trace_grid_alone: 1..16384"); c->tr_grid_alone = (int)value; return TIRT_OK; }
    if (!strcmp(name, "trace_grid")) { TIRT_REQUIRE(value >= 1 && value <= 16384, "trace_grid: 1..16384"); c->tr_grid = (int)value; c->grid_user = true; return TIRT_OK; }
    set_error(std::string("tirt_set_option: unknown option ") + name);
    return TIRT_ERR_ARG;
}

int tirt_scene_upload(tirt_ctx *c, const float *vertex, int nv, const int32_t *primitive, int n, const float *material, int nm,
                      const float *shape, int ns, const int32_t *light, int nl, int light_count, const float bmin[3], const float bmax[3])
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;      // scene data must not change under batches still in flight
    TIRT_REQUIRE(n >= 1 && nv >= 0 && nm >= 1 && ns >= 1 && nl >= 1, "tirt_scene_upload: need n,nm,ns,nl >= 1");
    TIRT_REQUIRE(vertex && primitive && material && shape && light && bmin && bmax, "tirt_scene_upload: null pointer");
    TIRT_REQUIRE(light_count >= 0 && light_count <= nl, "tirt_scene_upload: light_count out of range");
    TIRT_REQUIRE((long long)2 * n - 1 < (1ll << 24), "tirt_scene_upload: node indices are stored as f32 (exact below 2^24 nodes)");
    // validate indices on the host so that device code can trust them
    for (int i = 0; i < n; i++) {
        const int32_t *pr = primitive + (size_t)i * 3;
        if (pr[0] == PRIMITIVE_TRI) { TIRT_REQUIRE(pr[1] >= 0 && pr[1] + 2 < nv, "tirt_scene_upload: vertex index out of range"); }
        else { TIRT_REQUIRE(pr[1] >= 0 && pr[1] < ns, "tirt_scene_upload: shape index out of range"); }
        TIRT_REQUIRE(pr[2] >= 0 && pr[2] < nm, "tirt_scene_upload: material index out of range");
    }
    for (int i = 0; i < nl; i++) TIRT_REQUIRE(light[i] >= 0 && light[i] < n, "tirt_scene_upload: light index out of range");
    c->sphere_prims.clear();
    for (int i = 0; i < n; i++) {
        const int32_t *pr = primitive + (size_t)i * 3;
        if (pr[0] != PRIMITIVE_TRI && (int)shape[(size_t)pr[1] * 10] == SHAPE_SPHERE) c->sphere_prims.push_back(i);
    }
    hipStream_t st = c->stream;
    c->built = false;
    if (upload(c->vertex, vertex, sizeof(float) * 9 * (size_t)nv, st)) return TIRT_ERR_HIP;
    if (upload(c->primitive, primitive, sizeof(int) * 3 * (size_t)n, st)) return TIRT_ERR_HIP;
    if (upload(c->material, material, sizeof(float) * 10 * (size_t)nm, st)) return TIRT_ERR_HIP;
    if (upload(c->shape, shape, sizeof(float) * 10 * (size_t)ns, st)) return TIRT_ERR_HIP;
    if (upload(c->light, light, sizeof(int) * (size_t)nl, st)) return TIRT_ERR_HIP;
    c->nv = nv; c->n = n; c->nm = nm; c->ns = ns; c->nl = nl; c->light_count = light_count; c->shade_rec_valid = false;
    for (int k = 0; k < 3; k++) { c->bmin[k] = bmin[k]; c->bmax[k] = bmax[k]; }
    if (refresh_material_table(c)) return TIRT_ERR_HIP;
    if (!c->env.p) {        // default: 1x1 black (Scene.py:295-296 loads image/black.png)
        int32_t z = 0;
        if (upload(c->env, &z, sizeof(int32_t), st)) return TIRT_ERR_HIP;
        c->env_w = 1; c->env_h = 1; c->env_power = 0.0f;
    }
    TIRT_HIP(hipStreamSynchronize(st));
    return TIRT_OK;
}

int tirt_material_upload(tirt_ctx *c, const float *material, int nm)
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;      // scene data must not change under batches still in flight
    TIRT_REQUIRE(material && nm == c->nm, "tirt_material_upload: material count differs from the uploaded scene");
    if (upload(c->material, material, sizeof(float) * 10 * (size_t)nm, c->stream)) return TIRT_ERR_HIP;
    if (refresh_material_table(c)) return TIRT_ERR_HIP;
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_env_upload(tirt_ctx *c, const int32_t *rgb_packed, int w, int h, float power)
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;      // scene data must not change under batches still in flight
    TIRT_REQUIRE(rgb_packed && w >= 1 && h >= 1, "tirt_env_upload: bad image");
    if (upload(c->env, rgb_packed, sizeof(int32_t) * (size_t)w * h, c->stream)) return TIRT_ERR_HIP;
    c->env_w = w; c->env_h = h; c->env_power = power;
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_lbvh_build(tirt_ctx *c)
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;      // scene data must not change under batches still in flight
    return lbvh_build(c);
}

#ifdef TIRT_EXPERIMENTS
int tirt_exp_download(tirt_ctx *c, int which, void *out, uint64_t bytes)
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;
    DevBuf *b = which == 0 ? &c->tri : which == 1 ? &c->prim_slot : which == 2 ? &c->leaf_compact : which == 3 ? &c->cnode : which == 5 ? &c->bd[0].qidx : &c->cparent;
    TIRT_REQUIRE(bytes <= b->bytes, "tirt_exp_download: too many bytes");
    TIRT_HIP(hipMemcpy(out, b->p, bytes, hipMemcpyDeviceToHost));
    return TIRT_OK;
}
int tirt_exp_wide_from_tree(tirt_ctx *c, const float *compact_host, const int32_t *csize_host)
{
    CTX(c);
    TIRT_REQUIRE(c->built && c->n >= 2, "tirt_exp_wide_from_tree: LBVH not built");
    if (sync_all(c)) return TIRT_ERR_HIP;
    return exp_wide_from_tree(c, compact_host, csize_host);
}
#endif

int tirt_lbvh_download(tirt_ctx *c, int32_t *morton_sorted, float *bvh_node, float *compact_node)
{
    CTX(c);
    TIRT_REQUIRE(c->built, "tirt_lbvh_download: LBVH not built");
    const size_t n = c->n, N = 2 * n - 1;
    if (morton_sorted) TIRT_HIP(hipMemcpyAsync(morton_sorted, c->morton_sorted.p, sizeof(int) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    if (bvh_node) TIRT_HIP(hipMemcpyAsync(bvh_node, c->bvh_node.p, sizeof(float) * NOD_VEC * N, hipMemcpyDeviceToHost, c->stream));
    if (compact_node) TIRT_HIP(hipMemcpyAsync(compact_node, c->compact.p, sizeof(float) * CPN_VEC * N, hipMemcpyDeviceToHost, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_traversal_tree_download(tirt_ctx *c, float *rows)
{
    CTX(c);
    TIRT_REQUIRE(c->built && rows, "tirt_traversal_tree_download: LBVH not built / null pointer");
    const size_t N = 2 * (size_t)c->n - 1;
    const bool sah = c->built_sah != 0;
    TIRT_HIP(hipMemcpyAsync(rows, sah ? c->sah_compact.p : c->compact.p, sizeof(float) * N * CPN_VEC, hipMemcpyDeviceToHost, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_morton_download(tirt_ctx *c, int32_t *out)
{
    CTX(c);
    TIRT_REQUIRE(c->built && out, "tirt_morton_download: LBVH not built");
    TIRT_HIP(hipMemcpyAsync(out, c->morton_unsorted.p, sizeof(int) * 2 * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_process_normal(tirt_ctx *c, const int32_t *vertex_index)
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;      // scene data must not change under batches still in flight
    TIRT_REQUIRE(c->built && vertex_index, "tirt_process_normal: LBVH not built");
    if (c->nv == 0) return TIRT_OK;
    for (int i = 0; i < c->nv; i++) TIRT_REQUIRE(vertex_index[i] >= 0 && vertex_index[i] < c->n, "tirt_process_normal: vertex_index out of range");
    DevBuf vi, smooth;
    if (upload(vi, vertex_index, sizeof(int) * (size_t)c->nv, c->stream)) return TIRT_ERR_HIP;
    if (smooth.ensure(sizeof(float) * 3 * (size_t)c->nv)) { vi.release(); return TIRT_ERR_HIP; }
    const int B = 128, G = (c->nv + B - 1) / B;
    hipLaunchKernelGGL(k_smooth_normal, dim3(G), dim3(B), 0, c->stream, scene_view(c), c->nv, c->compact.as<float>(), vi.as<int>(), smooth.as<float>());
    hipLaunchKernelGGL(k_write_normal, dim3(G), dim3(B), 0, c->stream, c->vertex.as<float>(), c->nv, smooth.as<float>());
    c->shade_rec_valid = false;
    hipError_t e = hipStreamSynchronize(c->stream);
    vi.release(); smooth.release();
    TIRT_HIP(e);
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

int tirt_vertex_download(tirt_ctx *c, float *vertex)
{
    CTX(c);
    TIRT_REQUIRE(vertex, "tirt_vertex_download: null");
    TIRT_HIP(hipMemcpyAsync(vertex, c->vertex.p, sizeof(float) * 9 * (size_t)c->nv, hipMemcpyDeviceToHost, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_total_area(tirt_ctx *c, float *out)
{
    CTX(c);
    TIRT_REQUIRE(out && c->n >= 1, "tirt_total_area: no scene");
    DevBuf d;
    if (d.ensure(sizeof(float))) return TIRT_ERR_HIP;
    int cnt = c->light_count > 0 ? c->light_count : 1;      // the light field always has >= 1 entry (Scene.py:258-261)
    hipLaunchKernelGGL(k_total_area, dim3(1), dim3(64), 0, c->stream, scene_view(c), cnt, d.as<float>());
    hipError_t e = hipMemcpyAsync(out, d.p, sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    d.release();
    TIRT_HIP(e);
    return TIRT_OK;
}

int tirt_camera_set(tirt_ctx *c, const float view[16], const float view_inv[16], const float eye[3], float fx, float fy, float cx, float cy)
{
    CTX(c);
    TIRT_REQUIRE(view && view_inv && eye, "tirt_camera_set: null");
    memcpy(c->view, view, sizeof(float) * 16);
    memcpy(c->cam.view_inv, view_inv, sizeof(float) * 12);
    memcpy(c->cam.eye, eye, sizeof(float) * 3);
    c->cam.fx = fx; c->cam.fy = fy; c->cam.cx = cx; c->cam.cy = cy;
    c->cam_set = true;
    return TIRT_OK;
}

int tirt_film_create(tirt_ctx *c, int W, int H, int tile_rank, int tile_count, int tile_size)
{
    CTX(c);
    AFTER_RENDER(c);
    TIRT_REQUIRE(W >= 1 && H >= 1 && (long long)W * H < (1ll << 30), "tirt_film_create: bad size");
    TIRT_REQUIRE(tile_count >= 1 && tile_rank >= 0 && tile_rank < tile_count && tile_size >= 1, "tirt_film_create: bad tiling");
    const long NP = (long)W * H;
    if (c->hdr.ensure(sizeof(float) * 3 * (size_t)NP) || c->rgb.ensure(sizeof(float) * 3 * (size_t)NP)) return TIRT_ERR_HIP;
    c->W = W; c->H = H; c->tile_rank = tile_rank; c->tile_count = tile_count; c->tile_size = tile_size;
    c->tile_blocked = (H % 8 == 0 && tile_size % (8 * H) == 0 && ((long)W * H) % tile_size == 0) ? 1 : 0;       // local_to_pixel
    long ntiles = (NP + tile_size - 1) / tile_size, local = 0;
    for (long t = tile_rank; t < ntiles; t += tile_count) {
        long beg = t * tile_size, end = beg + tile_size; if (end > NP) end = NP;
        local += end - beg;
    }
    c->npix_local = local;
    c->bdpt_px.release();                        // BDPT vertex arrays belong to the film: start from zeros
    TIRT_HIP(hipMemsetAsync(c->hdr.p, 0, sizeof(float) * 3 * (size_t)NP, c->stream));
    TIRT_HIP(hipMemsetAsync(c->rgb.p, 0, sizeof(float) * 3 * (size_t)NP, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_film_clear(tirt_ctx *c)
{
    CTX(c);
    AFTER_RENDER(c);
    TIRT_REQUIRE(c->hdr.p, "tirt_film_clear: film not created");
    if (c->bdpt_px.p) TIRT_HIP(hipMemsetAsync(c->bdpt_px.p, 0, c->bdpt_px.bytes, c->stream));
    TIRT_HIP(hipMemsetAsync(c->hdr.p, 0, sizeof(float) * 3 * (size_t)c->W * c->H, c->stream));
    TIRT_HIP(hipMemsetAsync(c->rgb.p, 0, sizeof(float) * 3 * (size_t)c->W * c->H, c->stream));
    return TIRT_OK;
}

static int submit_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed, int max_depth, int stack_size, int flags, bool spectral)
{
    TIRT_REQUIRE(c->built, "render: LBVH not built");
    TIRT_REQUIRE(c->cam_set, "render: camera not set");
    TIRT_REQUIRE(c->hdr.p && c->npix_local >= 0, "render: film not created");
    TIRT_REQUIRE(frame_count >= 0 && max_depth >= 1 && max_depth <= 4096, "render: bad frame_count/max_depth");
    if (frame_count == 0) return TIRT_OK;
    auto &p = c->pend;
    if (p.valid && p.begin + (uint32_t)p.count == frame_begin && p.seed == seed && p.max_depth == max_depth &&
        p.stack_size == stack_size && p.flags == flags && p.spectral == spectral) {
        p.count += frame_count;
    } else {
        if (int rc = flush_pending(c)) return rc;
        p.valid = true; p.begin = frame_begin; p.count = frame_count; p.seed = seed;
        p.max_depth = max_depth; p.stack_size = stack_size; p.flags = flags; p.spectral = spectral;
    }
    if ((size_t)p.count * (size_t)(c->npix_local > 0 ? c->npix_local : 1) >= effective_merge_paths(c)) return flush_pending(c);
    return TIRT_OK;
}
int tirt_pt_rgb_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed, int max_depth, int stack_size, int flags)
{
    CTX_NOFLUSH(c);
    return submit_render(c, frame_begin, frame_count, seed, max_depth, stack_size, flags, false);
}
int tirt_pt_spec_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed, int max_depth, int stack_size, int flags)
{
    CTX_NOFLUSH(c);
    TIRT_REQUIRE(c->spec_set && c->spec_view, "tirt_pt_spec_render: spectral tables not uploaded (tirt_spectral_upload)");
    return submit_render(c, frame_begin, frame_count, seed, max_depth, stack_size, flags, true);
}

int tirt_bdpt_rgb_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed)
{
    CTX(c);
    return bdpt_render(c, frame_begin, frame_count, seed);
}
int tirt_bdpt_spec_render(tirt_ctx *c, uint32_t frame_begin, int frame_count, uint32_t seed)
{
    CTX(c);
    TIRT_REQUIRE(c->spec_set && c->spec_dev.p, "tirt_bdpt_spec_render: spectral tables not uploaded (tirt_spectral_upload)");
    return bdpt_render(c, frame_begin, frame_count, seed, true);
}

int tirt_tone_map(tirt_ctx *c, float exposure)
{
    CTX(c);
    AFTER_RENDER(c);
    TIRT_REQUIRE(c->hdr.p, "tirt_tone_map: film not created");
    long nvals = 3l * c->W * c->H;
    hipLaunchKernelGGL(k_tone_map, dim3((unsigned)((nvals + 255) / 256)), dim3(256), 0, c->stream, c->hdr.as<float>(), c->rgb.as<float>(), nvals, exposure);
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

int tirt_film_download(tirt_ctx *c, float *hdr, float *rgb)
{
    CTX(c);
    AFTER_RENDER(c);
    TIRT_REQUIRE(c->hdr.p, "tirt_film_download: film not created");
    size_t bytes = sizeof(float) * 3 * (size_t)c->W * c->H;
    if (hdr) TIRT_HIP(hipMemcpyAsync(hdr, c->hdr.p, bytes, hipMemcpyDeviceToHost, c->stream));
    if (rgb) TIRT_HIP(hipMemcpyAsync(rgb, c->rgb.p, bytes, hipMemcpyDeviceToHost, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    TIRT_HIP(hipGetLastError());
    return TIRT_OK;
}

int tirt_film_export_device(tirt_ctx *c, void *dev_dst)
{
    CTX(c);
    AFTER_RENDER(c);
    TIRT_REQUIRE(c->hdr.p && dev_dst, "tirt_film_export_device: film not created");
    TIRT_HIP(hipMemcpyAsync(dev_dst, c->hdr.p, sizeof(float) * 3 * (size_t)c->W * c->H, hipMemcpyDeviceToDevice, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_film_import_device(tirt_ctx *c, const void *dev_src)
{
    CTX(c);
    AFTER_RENDER(c);
    TIRT_REQUIRE(c->hdr.p && dev_src, "tirt_film_import_device: film not created");
    TIRT_HIP(hipMemcpyAsync(c->hdr.p, dev_src, sizeof(float) * 3 * (size_t)c->W * c->H, hipMemcpyDeviceToDevice, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    return TIRT_OK;
}

int tirt_trace_closest(tirt_ctx *c, const float *rays, int nr, int stack_size, int flags, float *out_hit, int32_t *out_prim, int32_t *counts)
{
    CTX(c);
    TIRT_REQUIRE(rays && out_hit && out_prim, "tirt_trace_closest: null");
    return launch_trace_batch(c, rays, nr, stack_size, flags, false, out_hit, out_prim, counts);
}

int tirt_trace_shadow(tirt_ctx *c, const float *rays, int nr, int stack_size, int flags, float *out_t, int32_t *out_prim, int32_t *counts)
{
    CTX(c);
    TIRT_REQUIRE(rays && out_t && out_prim, "tirt_trace_shadow: null");
    return launch_trace_batch(c, rays, nr, stack_size, flags, true, out_t, out_prim, counts);
}

int tirt_bvh_info(tirt_ctx *c, uint64_t out[4])
{
    CTX(c);
    TIRT_REQUIRE(out && c->built, "tirt_bvh_info: LBVH not built");
    const int nq = c->wide_nodes;
    out[0] = (uint64_t)nq * 64u; out[1] = (uint64_t)c->n * sizeof(float4) * TRI_STRIDE; out[2] = (uint64_t)nq;
    const int top = TR_TOP_SLOTS;
    out[3] = (uint64_t)(nq < top ? nq : top);
    return TIRT_OK;
}

int tirt_micro_gather_rate(tirt_ctx *c, uint64_t working_set_bytes, int iters, double *gbps_out)
{
    CTX(c);
    if (sync_all(c)) return TIRT_ERR_HIP;
    TIRT_REQUIRE(gbps_out && iters >= 1 && working_set_bytes >= 64 && working_set_bytes <= ((uint64_t)1 << 36), "tirt_micro_gather_rate: bad arguments");
    const uint32_t nrec = (uint32_t)(working_set_bytes / 64);
    const int blocks = 1536;
    DevBuf rec, out;
    if (rec.ensure((size_t)nrec * 64) || out.ensure(sizeof(float) * blocks * 256)) { rec.release(); out.release(); return TIRT_ERR_HIP; }
    hipError_t e = hipMemsetAsync(rec.p, 0, (size_t)nrec * 64, c->stream);
    float best = 1.0e30f;
    for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {          // first launch warms the caches
        (void)hipEventRecord(c->ev0, c->stream);
        hipLaunchKernelGGL(k_micro_gather, dim3(blocks), dim3(256), 0, c->stream, rec.as<float4>(), nrec, iters, out.as<float>());
        (void)hipEventRecord(c->ev1, c->stream);
        e = hipEventSynchronize(c->ev1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        if (rep > 0 && ms < best) best = ms;
    }
    rec.release(); out.release();
    TIRT_HIP(e);
    *gbps_out = (double)blocks * 256.0 * (double)iters * 64.0 / ((double)best * 1.0e-3) / 1.0e9;
    return TIRT_OK;
}

int tirt_stats(tirt_ctx *c, tirt_stats_t *out)
{
    CTX(c);
    TIRT_REQUIRE(out, "tirt_stats: null");
    if (ensure_counters(c)) return TIRT_ERR_HIP;
    DevCounters h;
    if (sync_all(c)) return TIRT_ERR_HIP;
    TIRT_HIP(hipMemcpyAsync(&h, c->dev_counters.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    drain_render_events(c);
    out->rays_closest = h.rays_closest; out->rays_shadow = h.rays_shadow;
    out->box_closest = h.box_closest; out->leaf_closest = h.leaf_closest;
    out->box_shadow = h.box_shadow; out->leaf_shadow = h.leaf_shadow;
    out->shaded = h.shaded; out->paths = h.paths; out->stack_overflow = h.stack_overflow;
    out->diag_it_node = h.it_node; out->diag_lanes_node = h.lanes_node; out->diag_it_leaf = h.it_leaf;
    out->diag_lanes_leaf = h.lanes_leaf; out->diag_refills = h.refills; out->diag_it_outer = h.it_outer;
    out->diag_wave_ticks = h.wave_ticks; out->diag_drain_ticks = h.drain_ticks; out->diag_waves = h.waves;
    out->ms_build = c->ms_build; out->ms_render = c->ms_render;
    out->ms_trace_closest = c->ms_trace_closest; out->ms_trace_shadow = c->ms_trace_shadow; out->ms_shade = c->ms_shade;
    out->launches_trace_closest = c->launches_trace_closest; out->launches_trace_shadow = c->launches_trace_shadow;
    out->launches_shade = c->launches_shade; out->launches_tail = 0;      // (the persistent tail kernel left the library in round 6; the field stays for the ABI)
    if (h.stack_overflow > 0) {      // results are wrong (subtrees were dropped): the statistics are filled in, the call reports it
        set_error("traversal stack overflow on " + std::to_string((unsigned long long)h.stack_overflow) + " rays: raise stack_size");
        // reported once: the counter is cleared, so that later tirt_stats calls of unrelated consumers do not keep failing
        TIRT_HIP(hipMemsetAsync((char *)c->dev_counters.p + offsetof(DevCounters, stack_overflow), 0, sizeof(h.stack_overflow), c->stream));
        TIRT_HIP(hipStreamSynchronize(c->stream));
        return TIRT_ERR_STACK;
    }
    return TIRT_OK;
}

static void drain_pvb_events(tirt_ctx *c)
{
    for (auto &pr : c->pvb_ev) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->pvb_build_ns += (unsigned long long)((double)ms * 1.0e6);
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    c->pvb_ev.clear();
    (void)hipGetLastError();
}

int tirt_primary_beam_stats(tirt_ctx *c, uint64_t out[12])
{
    TIRT_REQUIRE(c && out, "tirt_primary_beam_stats: null arguments");
    for (int k = 0; k < 12; k++) out[k] = 0;
    if (sync_all(c)) return TIRT_ERR_HIP;
    drain_pvb_events(c);
    out[5] = c->pvb_builds; out[6] = c->pvb_build_ns; out[7] = c->pvb_skipped;
    if (!c->pvb_stat.p) return TIRT_OK;
    unsigned long long h[12];
    TIRT_HIP(hipMemcpy(h, c->pvb_stat.p, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 5; k++) out[k] = h[k];
    for (int k = 8; k < 12; k++) out[k] = h[k];
    return TIRT_OK;
}

int tirt_trace_timeline(tirt_ctx *c, uint64_t *out, int max_waves, int *n_waves)
{
    CTX(c);
    TIRT_REQUIRE(out && n_waves && max_waves >= 0, "tirt_trace_timeline: null / negative arguments");
    if (sync_all(c)) return TIRT_ERR_HIP;
    const int n = c->timeline_waves < max_waves ? c->timeline_waves : max_waves;
    if (n > 0) TIRT_HIP(hipMemcpy(out, c->timeline.p, sizeof(uint64_t) * 4 * (size_t)n, hipMemcpyDeviceToHost));
    *n_waves = c->timeline_waves;
    return TIRT_OK;
}

int tirt_stats_reset(tirt_ctx *c)
{
    CTX(c);
    if (ensure_counters(c)) return TIRT_ERR_HIP;
    if (sync_all(c)) return TIRT_ERR_HIP;
    drain_render_events(c);
    TIRT_HIP(hipMemsetAsync(c->dev_counters.p, 0, sizeof(DevCounters), c->stream));
    TIRT_HIP(hipStreamSynchronize(c->stream));
    c->ms_render = c->ms_trace_closest = c->ms_trace_shadow = c->ms_shade = 0.0;
    c->launches_trace_closest = c->launches_trace_shadow = c->launches_shade = 0;
    drain_pvb_events(c); c->pvb_builds = c->pvb_build_ns = c->pvb_skipped = 0;
    return TIRT_OK;
}

int tirt_kat_math(tirt_ctx *c, int fn, const float *x, const float *y, float *out, int n)
{
    CTX(c);
    TIRT_REQUIRE(x && y && out && n >= 0, "tirt_kat_math: null");
    if (n == 0) return TIRT_OK;
    DevBuf dx, dy, dout;
    int rc = TIRT_OK;
    if (upload(dx, x, sizeof(float) * (size_t)n, c->stream) || upload(dy, y, sizeof(float) * (size_t)n, c->stream) || dout.ensure(sizeof(float) * (size_t)n)) rc = TIRT_ERR_HIP;
    if (rc == TIRT_OK) {
        hipLaunchKernelGGL(k_kat_math, dim3((n + 255) / 256), dim3(256), 0, c->stream, fn, dx.as<float>(), dy.as<float>(), dout.as<float>(), n);
        hipError_t e = hipMemcpyAsync(out, dout.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error(std::string("tirt_kat_math: ") + hipGetErrorString(e)); rc = TIRT_ERR_HIP; }
    }
    dx.release(); dy.release(); dout.release();
    return rc;
}

int tirt_kat_brdf(tirt_ctx *c, int which, const float *in, int in_stride, float *out, int out_stride, int n)
{
    CTX(c);
    TIRT_REQUIRE(in && out && n >= 0 && which >= 0 && which <= 18, "tirt_kat_brdf: bad args");
    const int need_in[19] = {19, 19, 17, 6, 2, 2, 2, 6, 3, 3, 3, 7, 2, 2, 2, 1, 8, 24, 12}, need_out[19] = {2, 3, 4, 3, 3, 2, 1, 3, 3, 3, 3, 4, 1, 1, 1, 1, 4, 3, 2};
    TIRT_REQUIRE(in_stride >= need_in[which] && out_stride >= need_out[which], "tirt_kat_brdf: stride too small");
    if (n == 0) return TIRT_OK;
    DevBuf din, dout;
    int rc = TIRT_OK;
    if (upload(din, in, sizeof(float) * (size_t)n * in_stride, c->stream) || dout.ensure(sizeof(float) * (size_t)n * out_stride)) rc = TIRT_ERR_HIP;
    if (rc == TIRT_OK) {
        hipLaunchKernelGGL(k_kat_brdf, dim3((n + 255) / 256), dim3(256), 0, c->stream, which, din.as<float>(), in_stride, dout.as<float>(), out_stride, n);
        hipError_t e = hipMemcpyAsync(out, dout.p, sizeof(float) * (size_t)n * out_stride, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { set_error(std::string("tirt_kat_brdf: ") + hipGetErrorString(e)); rc = TIRT_ERR_HIP; }
    }
    din.release(); dout.release();
    return rc;
}

}  // extern "C"
