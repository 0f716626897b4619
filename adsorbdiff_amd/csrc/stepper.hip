// Reverse-SDE/ODE stepper of the adsorbate placement, device resident.
//
// Reference: adsorbdiff/relaxation/diffusers/denoising_torch.py
//   :215-232  initial placement   com_xy <- cell . noise, z kept, rigid translation
//   :263-310  per step: per-system mean of the two heads over tag==2 atoms (f2 zeroed on fixed
//             atoms first, :498), ODE/SDE increment, z of the translation zeroed, wrap of the
//             COM into the cell via solve / mod / cell.f   (COLUMNS of `cell` act as lattice
//             vectors here, unlike the graph code — kept as is)
//   :312-320  cumulative early stop over *all* systems (allclose(dcom, 0, 1e-3, 1e-3))
//   :322-353  rigid rotation about the COM (axis-angle -> quaternion -> matrix,
//             utils/rot_utils.py:18-98) + translation of each adsorbate
// The reference does this with a Python loop over systems and a device->host sync per step; here
// it is two tiny kernels (one wave per system) and no host round trip.
#include "common.h"

struct StepParams {
    const float* cell;
    const int32_t* atom_offset;
    const int32_t* tags;
    const int32_t* fixed;
    float* pos;
    const float* f1;
    const float* f2;
    const float* z_tr;
    const float* z_rot;
    float* sys;  // [B][16]: com(3) dcom(3) drot(3) cnt conv
    int32_t* state;
    float* dcom_out;
    float* drot_out;
    int B;
    adf_step_coef c;
    const adf_step_coef* coefs_dev;  // optional schedule table on the device, indexed by state[4]
    int num_steps;
    int early_stop_count;
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float pymod1(float x) {
    // torch.remainder(x, 1): result has the sign of the divisor
    float r = fmodf(x, 1.0f);
    if (r < 0.f) r += 1.0f;
    return r;
}

// Solve A f = v, 3x3, partial pivoting (what LAPACK sgesv / torch.linalg.solve does).
__device__ void solve3(const float* A, const float* v, float* f) {
    float m[3][4];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) m[r][c] = A[3 * r + c]; m[r][3] = v[r]; }
    for (int col = 0; col < 3; ++col) {
        int piv = col;
        for (int r = col + 1; r < 3; ++r) if (fabsf(m[r][col]) > fabsf(m[piv][col])) piv = r;
        if (piv != col) for (int c = 0; c < 4; ++c) { float t = m[col][c]; m[col][c] = m[piv][c]; m[piv][c] = t; }
        for (int r = col + 1; r < 3; ++r) {
            const float l = m[r][col] / m[col][col];
            for (int c = col; c < 4; ++c) m[r][c] = m[r][c] - l * m[col][c];
        }
    }
    f[2] = m[2][3] / m[2][2];
    f[1] = (m[1][3] - m[1][2] * f[2]) / m[1][1];
    f[0] = (m[0][3] - m[0][1] * f[1] - m[0][2] * f[2]) / m[0][0];
}

__global__ __launch_bounds__(64) void adf_init_placement_kernel(const float* cell, const int32_t* atom_offset,
                                                                 const int32_t* tags, float* pos, const float* noise,
                                                                 int B) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int a0 = atom_offset[b], a1 = atom_offset[b + 1];
    float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
    for (int a = a0 + lane; a < a1; a += 64)
        if (tags[a] == 2) { sx += pos[3 * a]; sy += pos[3 * a + 1]; sz += pos[3 * a + 2]; cnt += 1.f; }
    sx = wsum(sx); sy = wsum(sy); sz = wsum(sz); cnt = wsum(cnt);
    const float inv = 1.0f / fmaxf(cnt, 1.0f);
    const float cx = sx * inv, cy = sy * inv, cz = sz * inv;
    const float* cl = cell + 9 * b;
    const float* nz = noise + 3 * b;
    // einsum("bi,bij->bj", noise, cell^T): out_j = sum_i noise_i * cell[j][i]
    const float nx = cl[0] * nz[0] + cl[1] * nz[1] + cl[2] * nz[2];
    const float ny = cl[3] * nz[0] + cl[4] * nz[1] + cl[5] * nz[2];
    (void)cz;
    for (int a = a0 + lane; a < a1; a += 64)
        if (tags[a] == 2) {
            pos[3 * a] = (pos[3 * a] - cx) + nx;
            pos[3 * a + 1] = (pos[3 * a + 1] - cy) + ny;
            // z: (z - com_z) + com_z, as the reference evaluates it
            pos[3 * a + 2] = (pos[3 * a + 2] - cz) + cz;
        }
}

// phase 1: per-system scores -> (dcom, drot), wrapped; per-system convergence AND-ed into state[2]
__global__ __launch_bounds__(64) void adf_step_reduce_kernel(StepParams p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int a0 = p.atom_offset[b], a1 = p.atom_offset[b + 1];
    float s[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = 0.f;
    for (int a = a0 + lane; a < a1; a += 64)
        if (p.tags[a] == 2) {
            const float keep = (p.fixed && p.fixed[a] == 1) ? 0.f : 1.f;
            s[0] += p.pos[3 * a]; s[1] += p.pos[3 * a + 1]; s[2] += p.pos[3 * a + 2];
            s[3] += p.f1[3 * a]; s[4] += p.f1[3 * a + 1]; s[5] += p.f1[3 * a + 2];
            s[6] += p.f2[3 * a] * keep; s[7] += p.f2[3 * a + 1] * keep; s[8] += p.f2[3 * a + 2] * keep;
            s[9] += 1.f;
        }
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = wsum(s[i]);
    if (lane != 0) return;
    const float cnt = fmaxf(s[9], 1.0f);
    // schedule scalars: by value, or (graph replay: identical launch every step) from the device table
    adf_step_coef c = p.c;
    if (p.coefs_dev) c = p.coefs_dev[min(p.state[4], p.num_steps - 1)];
    float com[3], dcom[3], drot[3];
    for (int k = 0; k < 3; ++k) {
        com[k] = s[k] / cnt;
        const float str = s[3 + k] / cnt;
        const float srot = s[6 + k] / cnt;
        float d = __fmul_rn(c.coef_tr, str);
        float r = __fmul_rn(__fmul_rn(__fmul_rn(c.rot_pre, srot), c.rot_dt), c.rot_g2);
        if (p.z_tr) d = __fadd_rn(d, __fmul_rn(c.noise_tr, p.z_tr[3 * b + k]));
        if (p.z_rot) r = __fadd_rn(r, __fmul_rn(c.noise_rot, p.z_rot[3 * b + k]));
        dcom[k] = d;
        drot[k] = r;
    }
    dcom[2] = 0.f;
    const float* cl = p.cell + 9 * b;
    float tgt[3] = {com[0] + dcom[0], com[1] + dcom[1], com[2] + dcom[2]};
    float fr[3];
    solve3(cl, tgt, fr);
    for (int k = 0; k < 3; ++k) fr[k] = pymod1(pymod1(fr[k]));
    bool conv = true;
    for (int j = 0; j < 3; ++j) {
        const float w = cl[3 * j] * fr[0] + cl[3 * j + 1] * fr[1] + cl[3 * j + 2] * fr[2];
        dcom[j] = w - com[j];
        conv = conv && (fabsf(dcom[j]) <= 1.0e-3f);
    }
    float* o = p.sys + 16 * b;
    for (int k = 0; k < 3; ++k) { o[k] = com[k]; o[3 + k] = dcom[k]; o[6 + k] = drot[k]; }
    if (!conv) atomicAnd(&p.state[2], 0);
    if (p.dcom_out) for (int k = 0; k < 3; ++k) p.dcom_out[3 * b + k] = dcom[k];
    if (p.drot_out) for (int k = 0; k < 3; ++k) p.drot_out[3 * b + k] = drot[k];
}

// phase 2: early-stop bookkeeping (block 0 decides, every block re-derives the same decision from
// state[] written by the *previous* launch's decision kernel) + rigid update of each adsorbate
__global__ void adf_step_decide_kernel(int32_t* state, int early_stop_count) {
    // state: [0]=cumulative converged count, [1]=frozen, [2]=all-converged flag of this step, [3]=steps applied,
    //        [4]=steps issued (index into the device schedule table)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (!state[1]) {
            if (state[2]) {
                state[0] += 1;
                if (early_stop_count > 0 && state[0] == early_stop_count) state[1] = 1;
            }
            if (!state[1]) state[3] += 1;
        }
        state[2] = 1;  // re-arm for the next step
        state[4] += 1;
    }
}

__global__ __launch_bounds__(64) void adf_step_apply_kernel(StepParams p) {
    if (p.state[1]) return;  // the reference's `break`: this and all later steps leave pos untouched
    const int b = blockIdx.x, lane = threadIdx.x;
    const int a0 = p.atom_offset[b], a1 = p.atom_offset[b + 1];
    const float* o = p.sys + 16 * b;
    const float cx = o[0], cy = o[1], cz = o[2];
    const float tx = o[3], ty = o[4], tz = o[5];
    const float ax = o[6], ay = o[7], az = o[8];
    // axis-angle -> quaternion (rot_utils.py:50-81)
    const float ang = sqrtf(ax * ax + ay * ay + az * az);
    const float half = 0.5f * ang;
    float k;
    if (fabsf(ang) < 1e-6f) k = 0.5f - (ang * ang) / 48.0f;
    else k = sinf(half) / ang;
    const float qr = cosf(half), qi = ax * k, qj = ay * k, qk = az * k;
    // quaternion -> matrix (rot_utils.py:18-47)
    const float two_s = 2.0f / (qr * qr + qi * qi + qj * qj + qk * qk);
    const float R00 = 1 - two_s * (qj * qj + qk * qk), R01 = two_s * (qi * qj - qk * qr), R02 = two_s * (qi * qk + qj * qr);
    const float R10 = two_s * (qi * qj + qk * qr), R11 = 1 - two_s * (qi * qi + qk * qk), R12 = two_s * (qj * qk - qi * qr);
    const float R20 = two_s * (qi * qk - qj * qr), R21 = two_s * (qj * qk + qi * qr), R22 = 1 - two_s * (qi * qi + qj * qj);
    for (int a = a0 + lane; a < a1; a += 64)
        if (p.tags[a] == 2) {
            const float rx = p.pos[3 * a] - cx, ry = p.pos[3 * a + 1] - cy, rz = p.pos[3 * a + 2] - cz;
            // (rel @ R^T) + dcom + com   (denoising_torch.py:331-336)
            p.pos[3 * a] = ((rx * R00 + ry * R01 + rz * R02) + tx) + cx;
            p.pos[3 * a + 1] = ((rx * R10 + ry * R11 + rz * R12) + ty) + cy;
            p.pos[3 * a + 2] = ((rx * R20 + ry * R21 + rz * R22) + tz) + cz;
        }
}

int32_t adf_stepper_init(const adf_batch* b, float* pos, const int32_t* tags, const float* noise,
                         hipStream_t s) {
    hipLaunchKernelGGL(adf_init_placement_kernel, dim3(b->num_systems), dim3(64), 0, s, b->cell, b->atom_offset, tags,
                       pos, noise, b->num_systems);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

int32_t adf_stepper_step(float* sys, const adf_batch* b, float* pos, const int32_t* tags, const int32_t* fixed,
                         const float* f1, const float* f2, const adf_step_coef* coef, const adf_step_coef* coefs_dev,
                         int num_steps, const float* z_tr, const float* z_rot, int32_t early_stop_count,
                         int32_t* state, float* dcom, float* drot, hipStream_t s) {
    StepParams p;
    p.cell = b->cell; p.atom_offset = b->atom_offset; p.tags = tags; p.fixed = fixed; p.pos = pos;
    p.f1 = f1; p.f2 = f2; p.z_tr = z_tr; p.z_rot = z_rot; p.sys = sys; p.state = state;
    p.dcom_out = dcom; p.drot_out = drot; p.B = b->num_systems; p.early_stop_count = early_stop_count;
    if (coef) p.c = *coef; else p.c = adf_step_coef{};
    p.coefs_dev = coefs_dev; p.num_steps = num_steps;
    hipLaunchKernelGGL(adf_step_reduce_kernel, dim3(p.B), dim3(64), 0, s, p);
    hipLaunchKernelGGL(adf_step_decide_kernel, dim3(1), dim3(64), 0, s, state, early_stop_count);
    hipLaunchKernelGGL(adf_step_apply_kernel, dim3(p.B), dim3(64), 0, s, p);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}

// Hand-off rule of the reference's final-frame -> LMDB converter (scripts/create_lmdbs/pred_traj_to_lmdb.py:81-90): if the
// lowest adsorbate atom (tag 2) is less than `min_gap` above the highest surface atom (tag 1), the whole adsorbate is
// lifted by |diff| + min_gap.  One wave per system, in place; lifted[b] (optional) = applied shift.
__global__ __launch_bounds__(64) void adf_lift_kernel(float* pos, const int32_t* tags, const int32_t* atom_offset,
                                                       float min_gap, float* lifted) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int a0 = atom_offset[b], a1 = atom_offset[b + 1];
    float zs = -3.0e38f, za = 3.0e38f;
    for (int a = a0 + lane; a < a1; a += 64) {
        const float z = pos[3 * a + 2];
        if (tags[a] == 1) zs = fmaxf(zs, z);
        if (tags[a] == 2) za = fminf(za, z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { zs = fmaxf(zs, __shfl_xor(zs, o)); za = fminf(za, __shfl_xor(za, o)); }
    float shift = 0.f;
    if (zs > -1.0e38f && za < 1.0e38f) {
        const float diff = za - zs;
        if (diff < min_gap) shift = fabsf(diff) + min_gap;
    }
    if (shift != 0.f)
        for (int a = a0 + lane; a < a1; a += 64)
            if (tags[a] == 2) pos[3 * a + 2] = pos[3 * a + 2] + shift;
    if (lifted && lane == 0) lifted[b] = shift;
}

extern "C" int32_t adf_lift_adsorbates(float* pos, const int32_t* tags, const int32_t* atom_offset, int32_t B, float min_gap,
                                       float* lifted, void* stream) {
    if (!pos || !tags || !atom_offset || B <= 0) { adf_set_error("lift_adsorbates: bad argument"); return ADF_EINVAL; }
    hipLaunchKernelGGL(adf_lift_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pos, tags, atom_offset, min_gap, lifted);
    ADF_HIP_CHECK(hipGetLastError());
    return ADF_OK;
}
