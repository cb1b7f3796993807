// lrc_sector.h -- the packet ("sector") trace kernel of the pose-batched scan of a multi-line sensor (gfx950 only).
// Included by lidarcast.hip inside its anonymous namespace, after TraceParams / write_back.
//
// What it replaces: for a sensor whose rays form a GRID in (scan line, azimuth) -- IndoorLidar with listed elevations,
// lidar/indoor_lidar.py:94-131: beta_i = az0 + i * az_step, one elevation per line -- the per-ray BVH descent of
// trace_kernel.  One wavefront owns a PACKET of rays: one pose x `nl` scan lines x 64 consecutive azimuths.  Instead of
// walking the tree once per ray (26 node steps per ray, 64 rays re-deriving what their neighbours already found, under
// 47 % lane utilisation), the wave walks the tree ONCE for the packet's frustum, one lane per NODE:
//   1. frustum traversal: a wave-shared stack in LDS, segmented by tree level; each round pops up to 64 nodes of the
//      deepest level (one per lane: 64 coalesced 64-byte node reads in flight), tests both child boxes against the four
//      planes that bound the packet's rays, and appends the survivors with ballot/popcount compaction -- inner nodes
//      to the stack, leaves to a leaf queue.
//   2. triangles, one lane per triangle: its bounding sphere against the packet in sensor angles gives the scan lines
//      it can touch (most 2 cm triangles fall BETWEEN the scan lines and die here) and a short azimuth interval.
//   3. candidates: for every (triangle, line) that survives, the few rays of that interval run the EXACT ray/triangle
//      test of the hit definition (tri_hit, lrc_device.h: same arithmetic as trace_kernel) and fold (t, triangle row) into
//      a 64-bit key per ray with an LDS atomic min -- unsigned order of (t bits, row) is the lexicographic order of the
//      definition.
//   4. the shared write-back (hit point, range filter, labels, ...) runs per ray as in trace_kernel.
// The result is the lexicographic minimum of (t, row) over all triangles tri_hit accepts, as long as every accepted
// (ray, triangle) pair is among the candidates.  Why it is: tri_hit accepts only if t lies in the padded slab interval of
// the triangle's own box, so the hit point o + t d lies within delta = 2.5e-4 t + 3e-5 of that box (the pads of
// slab_interval plus float32 rounding), hence within r_eff = r + sqrt(3) delta of the centre of the box's bounding
// sphere; the candidate cone (angular radius asin(r_eff / D) plus margins that dwarf the 1e-7 rounding of directions
// and the 1e-6 error of atan2f) therefore contains the ray, and every node box that contains the triangle, grown by
// the same delta, meets the packet's frustum.  Bit-identity with trace_kernel is asserted on every test scene.
#pragma once

struct SectorParams {
    TraceParams tp;                 // scene arrays, poses16, dirs3, options, outputs (epilogue = write_back<BY_PRIM>)
    const float4* slot_sphere;      // per leaf slot: centre of the triangle's box, radius of its bounding sphere
    uint32_t H, W;                  // scan lines, azimuths per line (W % 64 == 0)
    uint32_t nl;                    // scan lines per packet (1..8)
    uint32_t groups;                // ceil(H / nl)
    float az0, az_step;             // beta_i = az0 + i * az_step (radians)
    uint32_t stack_cap;             // entries of the level-segmented stack: 128 * (max_depth + 1)
    uint32_t levels;                // max_depth + 1
    uint64_t num_packets;
    unsigned long long* diag;       // DIAG build: totals over all packets (see kSectorDiagWords)
};

// DIAG counters: node rounds, nodes popped, leaves taken, triangles sphere-tested, pairs emitted, pair rounds,
// candidate rays tested, exact-test hits
constexpr int kSectorDiagWords = 8;

constexpr int kLeafQ = 256, kPairQ = 128;

struct alignas(8) SectorPair { uint32_t slot; uint32_t lo_hi_mask; };   // il_lo | il_hi << 8 | line mask << 16

// a wave-uniform float, moved to a scalar register
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// LDS layout (dynamic): keys[R] u64 | dtab[R*3] f32 | level_cnt[levels] u32 | stack[stack_cap] u32 | leafq[kLeafQ] u32 |
//                       pairq[kPairQ] SectorPair          with R = nl * 64
template <bool DIAG>
__global__ __launch_bounds__(64) void sector_kernel(const SectorParams q) {
    uint32_t dg[kSectorDiagWords] = {0, 0, 0, 0, 0, 0, 0, 0};
    extern __shared__ unsigned long long s_raw[];
    const TraceParams& p = q.tp;
    const uint32_t lane = threadIdx.x;
    const uint32_t R = q.nl * 64u;
    unsigned long long* s_keys = s_raw;
    float* s_dtab = (float*)(s_keys + R);
    uint32_t* s_lcnt = (uint32_t*)(s_dtab + R * 3);
    uint32_t* s_stack = s_lcnt + q.levels;
    uint32_t* s_leafq = s_stack + q.stack_cap;
    SectorPair* s_pairq = (SectorPair*)(s_leafq + kLeafQ);

    // ---- which packet ----
    const uint64_t pk = xcd_tile(blockIdx.x, gridDim.x);
    if (pk >= q.num_packets) return;
    const uint32_t azb = q.W / 64u;
    const uint64_t per_pose = (uint64_t)q.groups * azb;
    const uint64_t pose = pk / per_pose;
    const uint32_t rem = (uint32_t)(pk - pose * per_pose);
    const uint32_t grp = rem / azb, ab = rem - grp * azb;
    const uint32_t j0 = grp * q.nl, i0 = ab * 64u;
    const uint32_t nlines = (q.H - j0) < q.nl ? (q.H - j0) : q.nl;
    const uint64_t N = (uint64_t)q.H * q.W;
    const double* M = p.poses16 + pose * 16;
    const V3 o{uni((float)M[3]), uni((float)M[7]), uni((float)M[11])};
    // rotation, float32, for the culling tests only (rays are generated with the float64 chain of gen_ray)
    const float r00 = uni((float)M[0]), r01 = uni((float)M[1]), r02 = uni((float)M[2]);
    const float r10 = uni((float)M[4]), r11 = uni((float)M[5]), r12 = uni((float)M[6]);
    const float r20 = uni((float)M[8]), r21 = uni((float)M[9]), r22 = uni((float)M[10]);

    // ---- rays of the packet: keys, float32 world directions; per-line sin/cos of the elevation ----
    bool rays_finite = true;
    for (uint32_t jl = 0; jl < nlines; ++jl) {
        const uint32_t r = jl * 64u + lane;
        V3 oo, d;
        double cx, cy, cz;
        gen_ray(p.poses16, p.dirs3, pose, (uint64_t)(j0 + jl) * q.W + i0 + lane, oo, d, cx, cy, cz);
        s_dtab[r * 3] = d.x; s_dtab[r * 3 + 1] = d.y; s_dtab[r * 3 + 2] = d.z;
        s_keys[r] = ~0ull;
        rays_finite = rays_finite & finite_ray(oo, d);
    }
    // a non-finite pose or table entry anywhere in the packet: no culling geometry can be trusted -> every ray of the
    // packet is a miss only if ITS components are non-finite; handled by testing finiteness per candidate below and by
    // keeping the frustum test out of the way (all nodes pass) when the packet is not clean
    const bool clean = __builtin_amdgcn_ballot_w64(!rays_finite) == 0ull;

    // packet frustum in the sensor frame: azimuth wedge [b_lo, b_hi] and elevation band [tan_lo, tan_hi], widened
    float sin_l[8];
    float tan_lo = __builtin_inff(), tan_hi = -__builtin_inff();
    float cos_min = 1.0f;
#pragma unroll
    for (uint32_t jl = 0; jl < 8; ++jl) {
        sin_l[jl] = 4.0f;                                       // never within theta of a sine
        if (jl < nlines) {
            const double* dv = p.dirs3 + ((uint64_t)(j0 + jl) * q.W + i0) * 3;
            const double sz = dv[2], ch = __builtin_sqrt(dv[0] * dv[0] + dv[1] * dv[1]);
            sin_l[jl] = uni((float)sz);
            const float tn = (float)(sz / (ch > 1e-9 ? ch : 1e-9));
            tan_lo = min2(tan_lo, tn); tan_hi = max2(tan_hi, tn);
            cos_min = min2(cos_min, (float)ch);
        }
    }
    tan_lo = uni(tan_lo); tan_hi = uni(tan_hi); cos_min = uni(cos_min);
    const float b_a = q.az0 + (float)i0 * q.az_step, b_b = q.az0 + (float)(i0 + 63u) * q.az_step;
    const float b_c = 0.5f * (b_a + b_b);
    const float half = 0.5f * __builtin_fabsf(b_b - b_a) + 2e-4f;          // half width of the wedge, widened
    const float cb = __builtin_cosf(b_c), sb = __builtin_sinf(b_c);
    const float ch_ = __builtin_cosf(half), sh_ = __builtin_sinf(half);
    // wedge planes: inside <=> n.p >= 0.  Left/right boundaries are the centre direction turned by +-half.
    //   n_L = (-sin(b_c - half), cos(b_c - half), 0),  n_R = (sin(b_c + half), -cos(b_c + half), 0)
    const float sL = sb * ch_ - cb * sh_, cL = cb * ch_ + sb * sh_;        // sin / cos of b_c - half
    const float sR = sb * ch_ + cb * sh_, cR = cb * ch_ - sb * sh_;        // sin / cos of b_c + half
    // elevation planes: with q = x cos b_c + y sin b_c >= rho cos(half) inside the wedge,
    //   z <= tan_hi rho  =>  z <= T q with T = tan_hi / cos(half) (tan_hi >= 0) or T = tan_hi (tan_hi < 0)
    //   z >= tan_lo rho  =>  z >= B q with B = tan_lo / cos(half) (tan_lo <= 0) or B = tan_lo (tan_lo > 0)
    const float th = tan_hi + 3e-4f * (1.0f + tan_hi * tan_hi), tl = tan_lo - 3e-4f * (1.0f + tan_lo * tan_lo);
    const float Tt = th >= 0.0f ? th / ch_ : th, Bt = tl <= 0.0f ? tl / ch_ : tl;
    // sensor-frame normals -> world: n_w = R n_s (p_s = R^T (p_w - o))
    float pl[4][3];
    {
        const float ns[4][3] = {{-sL, cL, 0.0f}, {sR, -cR, 0.0f}, {Tt * cb, Tt * sb, -1.0f}, {-Bt * cb, -Bt * sb, 1.0f}};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pl[k][0] = uni(r00 * ns[k][0] + r01 * ns[k][1] + r02 * ns[k][2]);
            pl[k][1] = uni(r10 * ns[k][0] + r11 * ns[k][1] + r12 * ns[k][2]);
            pl[k][2] = uni(r20 * ns[k][0] + r21 * ns[k][1] + r22 * ns[k][2]);
        }
    }
    // a wedge wider than ~160 degrees is not bounded by two planes; elevations beyond ~88 degrees have no usable tangent
    const bool wedge_ok = clean && half < 1.4f && tan_hi < 40.0f && tan_lo > -40.0f;

    // does the box (lo, hi), grown by the slack of the hit definition, meet the packet's frustum?
    auto box_in = [&](float lox, float loy, float loz, float hix, float hiy, float hiz) -> bool {
        if (!wedge_ok) return true;
        const float cx = 0.5f * (lox + hix) - o.x, cy = 0.5f * (loy + hiy) - o.y, cz = 0.5f * (loz + hiz) - o.z;
        float hx = 0.5f * (hix - lox), hy = 0.5f * (hiy - loy), hz = 0.5f * (hiz - loz);
        const float ext = (__builtin_fabsf(cx) + __builtin_fabsf(cy) + __builtin_fabsf(cz)) + (hx + hy + hz);
        const float grow = fma_(ext, 6e-4f, 1e-4f);
        hx += grow; hy += grow; hz += grow;
        bool in = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float s = (pl[k][0] * cx + pl[k][1] * cy + pl[k][2] * cz) +
                            (__builtin_fabsf(pl[k][0]) * hx + __builtin_fabsf(pl[k][1]) * hy + __builtin_fabsf(pl[k][2]) * hz);
            in = in & (s >= 0.0f);
        }
        return in;
    };

    __syncthreads();

    // ---- wave-shared work lists (all counters are wave-uniform values every lane carries) ----
    uint32_t nleaf = 0, npair = 0;
    int cur = -1;                                   // deepest non-empty level of the stack
    uint32_t base_cur = 0;                          // first entry of level `cur`
    if (p.num_nodes) {
        if (lane < q.levels) s_lcnt[lane] = 0;
        if (lane == 0) { s_stack[0] = 0u; s_lcnt[0] = 1u; }
        cur = 0;
    }
    __syncthreads();

    // candidates of one (triangle, lines, azimuth interval): the exact test + atomic min into the ray keys
    auto run_pairs = [&](uint32_t m) {
        // lanes < m take the last m pairs
        SectorPair pr{0u, 0u};
        if (lane < m) pr = s_pairq[npair - 1u - lane];
        npair -= m;
        if (DIAG) dg[5] += 1;
        if (lane < m) {
            const uint32_t slot = pr.slot;
            const uint32_t il_lo = pr.lo_hi_mask & 0xFFu, il_hi = (pr.lo_hi_mask >> 8) & 0xFFu, lmask = pr.lo_hi_mask >> 16;
            const float4* tr = p.tris + (size_t)slot * 3;
            const float4 a = tr[0], b = tr[1], c = tr[2];
            const V3 v0{a.x, a.y, a.z}, e1{a.w, b.x, b.y}, e2{b.z, b.w, c.x}, ng{c.y, c.z, c.w};
            uint32_t prim = 0xFFFFFFFFu;
            for (uint32_t jl = 0; jl < nlines; ++jl) {
                if (!((lmask >> jl) & 1u)) continue;
                for (uint32_t il = il_lo; il <= il_hi; ++il) {
                    const uint32_t r = jl * 64u + il;
                    const V3 d{s_dtab[r * 3], s_dtab[r * 3 + 1], s_dtab[r * 3 + 2]};
                    if (!finite_ray(o, d)) continue;
                    float t;
                    if (DIAG) dg[6] += 1;
                    // the few pairs that pass the Moeller-Trumbore conditions form the ray's slab constants for the clause
                    if (tri_mt(o, d, v0, e1, e2, ng, t) &&
                        box_clause(make_slab(o, d), p.slot_box[(size_t)slot * 6], p.slot_box[(size_t)slot * 6 + 1],
                                   p.slot_box[(size_t)slot * 6 + 2], p.slot_box[(size_t)slot * 6 + 3],
                                   p.slot_box[(size_t)slot * 6 + 4], p.slot_box[(size_t)slot * 6 + 5], t)) {
                        if (DIAG) dg[7] += 1;
                        if (prim == 0xFFFFFFFFu) prim = p.slot_prim[slot];
                        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | prim;
                        atomicMin(&s_keys[r], key);
                    }
                }
            }
        }
        __syncthreads();
    };

    // one lane per leaf: every triangle of the leaf against the packet in sensor angles -> pairs
    auto run_leaves = [&](uint32_t m) {
        uint32_t enc = 0u;
        if (lane < m) enc = s_leafq[nleaf - 1u - lane];
        nleaf -= m;
        if (DIAG) dg[2] += lane < m ? 1 : 0;
        const uint32_t first = enc >> 3, cnt = enc & 7u;
        for (uint32_t k = 0; k < 4u; ++k) {
            if (__builtin_amdgcn_ballot_w64(k < cnt) == 0ull) break;
            if (npair > (uint32_t)kPairQ - 64u) run_pairs(64u);          // room for one pair per lane
            bool emit = false;
            SectorPair pr{0u, 0u};
            if (k < cnt) {
                if (DIAG) dg[3] += 1;
                const uint32_t slot = first + k;
                const float4 sp = q.slot_sphere[slot];
                const float vx = sp.x - o.x, vy = sp.y - o.y, vz = sp.z - o.z;
                // sensor frame: p = R^T v
                const float px = r00 * vx + r10 * vy + r20 * vz;
                const float py = r01 * vx + r11 * vy + r21 * vz;
                const float pz = r02 * vx + r12 * vy + r22 * vz;
                const float D2 = (px * px + py * py) + pz * pz;
                const float D = __builtin_sqrtf(D2);
                const float reff = fma_(sp.w, 1.002f, fma_(D, 5e-4f, 2e-4f));
                uint32_t lmask = 0u, il_lo = 0u, il_hi = 63u;
                const float x = reff / D;                               // sin of the cone's angular radius
                if (!(x < 0.45f) || !wedge_ok) {
                    lmask = (1u << nlines) - 1u;                         // near or huge: every ray of the packet
                } else {
                    const float theta = fma_(x * x, x, x) + 1e-4f;       // >= asin(x) for x < 0.45, plus margin
                    const float sc = pz / D;                            // sin of the centre's elevation
#pragma unroll
                    for (uint32_t jl = 0; jl < 8u; ++jl)
                        if (__builtin_fabsf(sin_l[jl] - sc) <= theta) lmask |= 1u << jl;
                    lmask &= (1u << nlines) - 1u;
                    if (lmask) {
                        const float rho = __builtin_sqrtf(px * px + py * py);
                        const float cc = rho / D;                       // cos of the centre's elevation
                        const float y = theta / max2(cc, cos_min);      // sin of the azimuth half width (see header)
                        if (y < 0.45f) {
                            const float dbeta = fma_(y * y, y, y) + 1e-4f;
                            const float beta = atan2f(py, px);
                            float rel = (beta - q.az0) / q.az_step - (float)i0;     // real-valued index inside the packet
                            const float Wf = (float)q.W;
                            rel -= Wf * __builtin_rintf(rel / Wf);                   // wrap to [-W/2, W/2]
                            const float di = dbeta / __builtin_fabsf(q.az_step) + 0.05f;
                            const float flo = __builtin_ceilf(rel - di), fhi = __builtin_floorf(rel + di);
                            // the packet may also be reached across the seam: try the other wrap when out of range
                            float lo2 = flo, hi2 = fhi;
                            if (fhi < 0.0f) { lo2 = flo + Wf; hi2 = fhi + Wf; }
                            else if (flo > 63.0f) { lo2 = flo - Wf; hi2 = fhi - Wf; }
                            if (hi2 < 0.0f || lo2 > 63.0f) lmask = 0u;
                            else {
                                il_lo = (uint32_t)max2(lo2, 0.0f);
                                il_hi = (uint32_t)min2(hi2, 63.0f);
                            }
                        }
                    }
                }
                emit = lmask != 0u;
                pr.slot = slot;
                pr.lo_hi_mask = il_lo | (il_hi << 8) | (lmask << 16);
            }
            if (DIAG) dg[4] += emit ? 1 : 0;
            const unsigned long long bm = __builtin_amdgcn_ballot_w64(emit);
            if (emit) s_pairq[npair + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = pr;
            npair += (uint32_t)__popcll(bm);
            __syncthreads();
        }
    };

    // ---- main loop ----
    while (true) {
        if (npair >= 64u) { run_pairs(64u); continue; }
        if (nleaf >= 64u) { run_leaves(64u); continue; }
        if (cur < 0) {                                    // tree exhausted: drain what is queued
            if (nleaf) { run_leaves(nleaf); continue; }
            if (npair) { run_pairs(npair); continue; }
            break;
        }
        // a round of nodes: up to 64 of the deepest level
        uint32_t cnt_cur = s_lcnt[cur];
        if (cnt_cur == 0u) {
            --cur;
            if (cur >= 0) base_cur -= s_lcnt[cur];
            continue;
        }
        const uint32_t m = cnt_cur < 64u ? cnt_cur : 64u;
        if (DIAG) { dg[0] += lane == 0 ? 1 : 0; dg[1] += lane < m ? 1 : 0; }
        int ref = -1;
        if (lane < m) ref = (int)s_stack[base_cur + cnt_cur - 1u - lane];
        bool in0 = false, in1 = false;
        int c0 = ~0, c1 = ~0;
        if (lane < m) {
            const F4* n = (const F4*)(p.nodes + (size_t)ref * 4);
            const F4 q0 = n[0], q1 = n[1], q2 = n[2], q3 = n[3];
            c0 = __float_as_int(q3.x); c1 = __float_as_int(q3.y);
            in0 = (c0 != ~0) && box_in(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y);
            in1 = (c1 != ~0) && box_in(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w);
        }
        // survivors: inner children -> next level of the stack, leaves -> leaf queue (ballot compaction)
        const uint32_t nxt_base = base_cur + (cnt_cur - m);     // level cur shrinks by m; level cur+1 starts after it
        const bool i0_ = in0 & (c0 >= 0), i1_ = in1 & (c1 >= 0);
        const bool l0_ = in0 & (c0 < 0), l1_ = in1 & (c1 < 0);
        const unsigned long long bi0 = __builtin_amdgcn_ballot_w64(i0_), bi1 = __builtin_amdgcn_ballot_w64(i1_);
        const unsigned long long bl0 = __builtin_amdgcn_ballot_w64(l0_), bl1 = __builtin_amdgcn_ballot_w64(l1_);
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t ni0 = (uint32_t)__popcll(bi0), ni = ni0 + (uint32_t)__popcll(bi1);
        const uint32_t nl0 = (uint32_t)__popcll(bl0), nlv = nl0 + (uint32_t)__popcll(bl1);
        if (i0_) s_stack[nxt_base + (uint32_t)__popcll(bi0 & below)] = (uint32_t)c0;
        if (i1_) s_stack[nxt_base + ni0 + (uint32_t)__popcll(bi1 & below)] = (uint32_t)c1;
        if (l0_) s_leafq[nleaf + (uint32_t)__popcll(bl0 & below)] = (uint32_t)(~c0);
        if (l1_) s_leafq[nleaf + nl0 + (uint32_t)__popcll(bl1 & below)] = (uint32_t)(~c1);
        nleaf += nlv;
        if (lane == 0) {
            s_lcnt[cur] = cnt_cur - m;
            if (ni) s_lcnt[cur + 1] = ni;
        }
        if (ni) { base_cur = nxt_base; ++cur; }
        __syncthreads();
    }

    if (DIAG) {
        if (q.diag) {
            if (lane != 0) { dg[5] = 0; }
#pragma unroll
            for (int k = 0; k < kSectorDiagWords; ++k) {
                unsigned int v = dg[k];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) atomicAdd(&q.diag[k], (unsigned long long)v);
            }
        }
    }
    // ---- write-back, one pass per scan line of the packet ----
    for (uint32_t jl = 0; jl < nlines; ++jl) {
        const uint64_t gid = pose * N + (uint64_t)(j0 + jl) * q.W + i0 + lane;
        V3 oo, d;
        double cx, cy, cz;
        gen_ray(p.poses16, p.dirs3, pose, (uint64_t)(j0 + jl) * q.W + i0 + lane, oo, d, cx, cy, cz);
        const unsigned long long key = s_keys[jl * 64u + lane];
        const float tb = key == ~0ull ? __builtin_inff() : __uint_as_float((uint32_t)(key >> 32));
        const uint32_t prim = key == ~0ull ? 0xFFFFFFFFu : (uint32_t)key;
        write_back<true, true>(p, gid, lane, oo, d, cx, cy, cz, tb, prim);
    }
}
