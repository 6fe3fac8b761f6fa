// fw_render: the FPV image of the analytic scene (sphere duck, cylinder obstacles, ground plane, sky) that a14's image
// functionals are computed on, written out pixel by pixel for a network to consume -- what Camera.capture_image() hands the
// env in the reference (segImg / depthImg, envs/fixedwing_objlock_env.py:603-622) and what its CNN path feeds a detector
// (envs/fixedwing_envs/objlock_yolo_env.py:646-716: a network's mask replaces segImg).
//   out[env][0][y][x] = 1.0 where the pixel's ray hits the duck sphere between the clip planes (none if a cylinder blocks the
//                       line of sight to its centre: the scene's binary occlusion rule), else 0.0
//   out[env][1][y][x] = depth-buffer value in [0, 1] of the nearest fragment (duck pixels: the sphere; others: ground or
//                       cylinder; sky = 1.0), far (t - near) / (t (far - near)) with t clipped to [near, far]
// at `res` x `res` pixels of the same body-fixed camera (FOV, tilt, offset of fw_config; the focal length scales with the
// width), for the env's CURRENT pose.  One workgroup (four waves) per env; two waves share the set-up, then every wave takes
// strips of 4 columns.
//   * The duck MASK is an exact comparison against 0 at the silhouette, so the test asks for the same bits, not for a tolerance:
//     the pose, the duck in the camera frame and the silhouette discriminant are the CPU checker's expressions statement by
//     statement in double, no FMA contraction (multiplications and additions only: cheap); the clip-plane test of a duck pixel
//     takes the fast quotient below and falls back to the IEEE sequences inside 1e-9 of a clip plane.
//   * The DEPTH channel is a float32 within 1e-7 of the checker's, not the same bits: it is computed on 1 / t -- the ray direction
//     as an affine form of the pixel coordinates (Hf + a Hr + b Hd: 6 fused multiply-adds instead of 27 operations), the ground's
//     1 / t a multiple of the direction's z (no division), a cylinder's from v_rsq / v_rcp seeds with Newton steps (~1 ulp of
//     double), the depth-buffer value c1 (1 - near / t) -- which took the vector instructions of a launch from 23.1 M to a third.
//     A cylinder's edge or top can fall on the other side of a pixel centre than in the checker only inside ~1e-15 of it.
//   * Round 3 tested every pixel against every cylinder, square root and division included: 818 vector instructions per
//     pixel with 20 cylinders, 70 % of the chip's fp64 issue peak -- the kernel was compute-bound at 3 % of the HBM write rate.
//     A ray hits an (infinite) cylinder iff its horizontal direction lies in the wedge between the two vertical tangent planes
//     through the camera, and the ray direction is affine in the pixel coordinates: the wedge is two half-planes of the image.
//     Lane o of the first wave builds them for cylinder o (plus the line-of-sight occlusion test of the duck, one cylinder per
//     lane instead of a serial loop on one thread); a wave then evaluates, for its tile, both affine forms at the four corner
//     pixels for every cylinder at once (lane = cylinder x corner) -- an affine form that is negative at all four corners of a
//     rectangle is negative inside -- and a pixel loops over the surviving cylinders only (bit mask, wave-uniform).  The margin
//     of the cull is seven orders of magnitude above rounding; what survives is decided by the exact expressions.
//   * the set-up waves leave pose, duck and cylinder tables in LDS; behind the barrier every wave moves the wave-uniform ones into
//     scalar registers (v_readfirstlane), and strip, row and cylinder loops are scalar control flow.
#pragma once
#include "fwsim_device.hpp"
#include "fwsim_objlock.hpp"

namespace fwsim {

struct RenderC {            // camera constants in double (built on the host from fw_config)
  double cam_f[3], cam_r[3], cam_d[3], cam_off[3];
  double tan_half_fov, near_, far_, duck_radius, obst_radius;
  double inv_near, inv_far, db_c1;      // 1 / near, 1 / far, far / (far - near): the depth-buffer value of a fragment at depth t is db_c1 (1 - near / t)
};

// A wave's tile: a strip of 4 columns x 64 rows, four pixels per lane (column x0 + lane % 4, rows y0 + lane / 4 + 16 k).  Cylinders are
// vertical and the camera rolls little: what a cylinder covers is a band of columns, and a narrow strip is crossed by few of them
// (16 x 16 tiles kept ~4 of 20 cylinders alive per tile, every one of them ~40 vector instructions for every pixel of the tile).
constexpr int kRTileW = 4, kRTileH = 64;

// A wave-uniform double moved into scalar registers (two v_readfirstlane): what every lane read from the same LDS word needs no
// vector register per lane, and a vector instruction takes one scalar operand for free.
__device__ __forceinline__ double render_uniform(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}
__device__ __forceinline__ double render_from_lane(double v, int src_lane) {      // lane `src_lane`'s value, in scalar registers
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane);
  return u.d;
}
// Four gathers from one scalar base with 32-bit byte offsets, issued back to back and waited for once.  Written out because hipcc,
// given the four loads as C++, waits for the first before it has even formed the addresses of the others (two round trips, not one).
__device__ __forceinline__ void render_gather4(const double* base, unsigned oa, unsigned ox, unsigned oy, unsigned oh, double& a, double& x, double& y, double& h) {
  asm volatile("global_load_dwordx2 %0, %4, %8\n\tglobal_load_dwordx2 %1, %5, %8\n\tglobal_load_dwordx2 %2, %6, %8\n\tglobal_load_dwordx2 %3, %7, %8\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(x), "=&v"(y), "=&v"(h) : "v"(oa), "v"(ox), "v"(oy), "v"(oh), "s"(base));
}
__device__ __forceinline__ void render_gather4(const float* base, unsigned oa, unsigned ox, unsigned oy, unsigned oh, float& a, float& x, float& y, float& h) {
  asm volatile("global_load_dword %0, %4, %8\n\tglobal_load_dword %1, %5, %8\n\tglobal_load_dword %2, %6, %8\n\tglobal_load_dword %3, %7, %8\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(x), "=&v"(y), "=&v"(h) : "v"(oa), "v"(ox), "v"(oy), "v"(oh), "s"(base));
}

// Dev-only (-DFW_RENDER_PROF, tools/render_wave_profile.sh): lane 0 of every wave leaves the cycle counter at six points of its life
#ifdef FW_RENDER_PROF
__device__ long long* g_render_prof = nullptr;      // [n_envs][4 waves][8]
#define FW_RP(i) do { if (lane == 0 && g_render_prof) g_render_prof[((size_t)env * 4 + wave) * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FW_RP(i) do { } while (0)
#endif

template <typename T, bool STAGE>
__global__ __launch_bounds__(256) void fw_render_kernel(const T* __restrict__ r, int tile, int n_envs, int nwaves, RenderC K, int res,
                                                        float* __restrict__ out, int stage_px) {
#pragma clang fp contract(off)                    // this kernel only: multiply-adds stay two roundings, as in the CPU checker's C
  __shared__ double s_pose[28];                   // R[9] cam[3] zc xc yc k2 | Hf Hr Hd (the camera axes in the world frame)
  __shared__ float s_dbox[4];                     // image-plane box that holds the duck's silhouette: a min / max, b min / max (culling only)
  __shared__ double s_cyl[FW_MAX_OBSTACLES][4];   // ox, oy, cc, height: cam - axis (horizontal), |.|^2 - radius^2
  __shared__ float s_wedge[FW_MAX_OBSTACLES][2][3];   // the two tangent half-planes as affine forms g(a, b) = g0 + a g1 + b g2 (>= 0 inside); [.][0][0] = +inf: never cull
  __shared__ float s_margin[FW_MAX_OBSTACLES];
  __shared__ int s_blocked, s_nob;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* s_ab = reinterpret_cast<double*>(smem_raw);      // [res]: (i - u0) / F, the image-plane coordinate of pixel column / row i (the image is square)
  const int env = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // (nwaves = blockDim.x / 64, handed in: a scalar the compiler knows is uniform, no fetch from the dispatch packet)
  if (env >= n_envs) return;
  const int setup_wave = env & (nwaves - 1), cyl_wave = (setup_wave + 1) & (nwaves - 1);
  const bool do_pose = wave == setup_wave, do_cyl = wave == cyl_wave;
  FW_RP(0);
  T gA = (T)0, gX = (T)0, gY = (T)0, gH = (T)0;
  if (do_pose || do_cyl) {                          // (first thing in the kernel: the fetch is the longest single wait of a set-up wave, 3.3 k of its 17.4 k cycles)
    // Everything the set-up reads comes in ONE round trip: four lane-indexed gathers issued back to back (lane l of the first fetches
    // the l-th of quaternion 4, position 3, duck 3, cylinder count; lane o of the others cylinder o's x / y / height -- the slots exist
    // whatever the count), handed out by v_readlane.  As scalar loads behind their uses the same words were four round trips in a
    // row (count, quaternion, position + duck, cylinders), and a workgroup is idle until they are in: without its pixel arithmetic this
    // kernel took 18.6 of its 27.4 us (4096 x 32 x 32)
    const T* rb = r + tile_index(tile, RF_COUNT, 0, env);
    const int fA = lane < 4 ? RF_QUAT + lane : lane < 7 ? RF_POS + (lane - 4) : lane < 10 ? RF_TASK + FW_ST_DUCK_POS + (lane - 7) : RF_TASK + FW_ST_NUM_OBST;
    const int lo = lane < FW_MAX_OBSTACLES ? lane : 0;
    const unsigned fstride = (unsigned)tile * (unsigned)sizeof(T), oX = (unsigned)(RF_TASK + FW_ST_OBST + 3 * lo) * fstride;       // (a tile of <= 64 envs x 171 fields: 32-bit offsets)
    render_gather4(rb, (unsigned)fA * fstride, oX, oX + fstride, oX + 2u * fstride, gA, gX, gY, gH);
    FW_RP(1);
  }
  const double W = (double)res, F = 0.5 * W / K.tan_half_fov, u0 = 0.5 * (W - 1.0), near = K.near_, far = K.far_;
  // (one IEEE division per column, not four per pixel; the wave BEHIND the two set-up waves starts the table, so the three run side by side)
  for (int i = ((wave - setup_wave - 2) & (nwaves - 1)) * 64 + lane; i < res; i += nwaves * 64) s_ab[i] = ((double)i - u0) / F;
  // Two waves share the set-up (one, with 64 threads per env): both fetch the state and rotate the camera, `setup_wave` goes on to the
  // duck and the ray basis, `cyl_wave` to the cylinders (occlusion of the duck, tangent wedges) -- the wave profile put 4.3 k of a
  // wave's 17.4 k cycles into this arithmetic, one dependent chain with three waves waiting for it
  if (do_pose || do_cyl) {
    // the pose, the duck in the camera frame, the ray basis: every lane of ONE wave for itself (no broadcast inside the wave), lane 0
    // leaves them for the other waves.  Which wave rotates with the env: a workgroup's wave i runs on SIMD i, and with the pixel loop
    // as short as it now is the set-up would otherwise queue sixteen deep on SIMD 0 of every CU.
    auto fld = [&](int l) { return render_from_lane((double)gA, l); };
    int nob = (int)fld(10);
    nob = nob < 0 ? 0 : (nob > FW_MAX_OBSTACLES ? FW_MAX_OBSTACLES : nob);
    double R[9], cam[3], relw[3], zc, xc, yc, k2, Hf[3], Hr[3], Hd[3];
    const double x = fld(0), y = fld(1), z = fld(2), w = fld(3);
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;                 // btMatrix3x3::setRotation
    const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs;
    const double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy; R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
    for (int k = 0; k < 3; ++k) cam[k] = fld(4 + k) + (R[3 * k] * K.cam_off[0] + R[3 * k + 1] * K.cam_off[1] + R[3 * k + 2] * K.cam_off[2]);
    const double Rd = K.duck_radius;
    const double C[3] = { fld(7), fld(8), fld(9) + Rd };
    for (int k = 0; k < 3; ++k) relw[k] = C[k] - cam[k];
    double relb[3];
    for (int k = 0; k < 3; ++k) relb[k] = R[k] * relw[0] + R[3 + k] * relw[1] + R[6 + k] * relw[2];       // R^T relw
    zc = relb[0] * K.cam_f[0] + relb[1] * K.cam_f[1] + relb[2] * K.cam_f[2];
    xc = relb[0] * K.cam_r[0] + relb[1] * K.cam_r[1] + relb[2] * K.cam_r[2];
    yc = relb[0] * K.cam_d[0] + relb[1] * K.cam_d[1] + relb[2] * K.cam_d[2];
    k2 = zc * zc + xc * xc + yc * yc - Rd * Rd;
    for (int k = 0; k < 3; ++k) {                    // (culling only)
      Hf[k] = R[3 * k] * K.cam_f[0] + R[3 * k + 1] * K.cam_f[1] + R[3 * k + 2] * K.cam_f[2];
      Hr[k] = R[3 * k] * K.cam_r[0] + R[3 * k + 1] * K.cam_r[1] + R[3 * k + 2] * K.cam_r[2];
      Hd[k] = R[3 * k] * K.cam_d[0] + R[3 * k + 1] * K.cam_d[1] + R[3 * k + 2] * K.cam_d[2];
    }
    if (do_pose && lane == 0) {
      for (int k = 0; k < 9; ++k) s_pose[k] = R[k];
      for (int k = 0; k < 3; ++k) s_pose[9 + k] = cam[k];
      s_pose[12] = zc; s_pose[13] = xc; s_pose[14] = yc; s_pose[15] = k2;
      for (int k = 0; k < 3; ++k) { s_pose[16 + k] = Hf[k]; s_pose[19 + k] = Hr[k]; s_pose[22 + k] = Hd[k]; }
      s_pose[25] = cam[2] > 0.0 ? -M<double>::div_(1.0, cam[2]) : 0.0;      // (depth only: ~1 ulp, not the IEEE sequence)
      // the silhouette's box (culling only): a point C + Rd u of the sphere projects to a = (xc + Rd ux) / (zc + Rd uz), which is off
      // xc / zc by |Rd (ux zc - uz xc)| / (zc (zc + Rd uz)) <= Rd (zc + |xc|) / (zc (zc - Rd)) -- asked only where zc - Rd > near > 0
      // (duck_possible below); the margin is orders of magnitude above the float conversions of the box and of the pixel coordinates
      const float zf = (float)zc, xf = (float)xc, yf = (float)yc, rf = (float)Rd, zs_ = fmaxf(zf - rf, 1e-9f), izz = __builtin_amdgcn_rcpf(zf * zs_);
      const float ca = xf * zs_ * izz, cb = yf * zs_ * izz, wa = rf * (zf + fabsf(xf)) * izz, wb = rf * (zf + fabsf(yf)) * izz;
      const float ma = 1e-5f * (1.0f + fabsf(ca) + wa), mb = 1e-5f * (1.0f + fabsf(cb) + wb);
      s_dbox[0] = ca - wa - ma; s_dbox[1] = ca + wa + ma; s_dbox[2] = cb - wb - mb; s_dbox[3] = cb + wb + mb;
    }
    bool blocked_any = false;
    if (do_cyl && lane < FW_MAX_OBSTACLES) {         // lane o: cylinder o
      bool blocked = false;
      double ox = 0.0, oy = 0.0, cc = 1.0, hh = 0.0;
      if (lane < nob) {
        const double ax = (double)gX, ay = (double)gY;
        hh = (double)gH;
        ox = cam[0] - ax; oy = cam[1] - ay;
        cc = ox * ox + oy * oy - K.obst_radius * K.obst_radius;
        // occluded(): the segment cam -> duck centre against this cylinder
        const double a = relw[0] * relw[0] + relw[1] * relw[1], b = 2.0 * (ox * relw[0] + oy * relw[1]);
        if (a > 0.0) {
          const double disc = b * b - 4.0 * a * cc;
          if (disc >= 0.0) {
            const double tt = (-b - ::sqrt(disc)) / (2.0 * a);
            if (tt > 0.0 && tt < 1.0) { const double zz2 = cam[2] + tt * relw[2]; if (zz2 >= 0.0 && zz2 <= hh) blocked = true; }
          }
        }
      }
      s_cyl[lane][0] = ox; s_cyl[lane][1] = oy; s_cyl[lane][2] = cc; s_cyl[lane][3] = hh;
      // the wedge (culling only; float): with e = axis - cam = (-ox, -oy), a horizontal direction h is inside iff
      // tan(beta) (h . e) -+ (h x e) >= 0, sin(beta) = radius / |e|, tan(beta) = radius / sqrt(cc)
      float g[2][3] = {{__builtin_inff(), 0.f, 0.f}, {0.f, 0.f, 0.f}}, margin = 0.f;
      if (lane < nob && cc > 1e-9) {
        const float ex = (float)-ox, ey = (float)-oy, tb = (float)K.obst_radius * __builtin_amdgcn_rsqf((float)cc);
        const float H[3][2] = {{(float)Hf[0], (float)Hf[1]}, {(float)Hr[0], (float)Hr[1]}, {(float)Hd[0], (float)Hd[1]}};
        for (int k = 0; k < 3; ++k) {
          const float dot = H[k][0] * ex + H[k][1] * ey, crs = H[k][0] * ey - H[k][1] * ex;
          g[0][k] = tb * dot - crs; g[1][k] = tb * dot + crs;
        }
        // seven orders of magnitude above float rounding, relative to the forms' scale |h| |e| (1 + tan beta), |h| <= 1 + |a| + |b|
        margin = 1e-4f * (1.0f + tb) * sqrtf(ex * ex + ey * ey) * (1.0f + 2.0f * (float)(K.tan_half_fov > 1.0 ? K.tan_half_fov : 1.0));
      }
      for (int k = 0; k < 3; ++k) { s_wedge[lane][0][k] = g[0][k]; s_wedge[lane][1][k] = g[1][k]; }
      s_margin[lane] = margin;
      blocked_any = blocked;
    }
    blocked_any = __any(blocked_any);
    if (do_cyl && lane == 0) { s_blocked = blocked_any ? 1 : 0; s_nob = nob; }
  }
  FW_RP(2);
  __syncthreads();
  FW_RP(3);
  const int nob = __builtin_amdgcn_readfirstlane(s_nob);
  // the pose constants are the same in every lane: all but one operand of each expression below live in scalar registers (20 vector
  // registers fewer: eight waves per SIMD instead of six or seven)
  double cam[3], Hf[3], Hr[3], Hd[3];
  for (int k = 0; k < 3; ++k) { Hf[k] = s_pose[16 + k]; Hr[k] = render_uniform(s_pose[19 + k]); Hd[k] = render_uniform(s_pose[22 + k]); }
  cam[2] = render_uniform(s_pose[11]);
  const double zc = s_pose[12], xc = render_uniform(s_pose[13]), yc = render_uniform(s_pose[14]), k2 = render_uniform(s_pose[15]);
  const double inv_near = K.inv_near, inv_far = K.inv_far, db_c1 = K.db_c1;
  const double kground = render_uniform(s_pose[25]);                                                 // -1 / (camera height); 0 for a camera at or below the ground (it sees none: 1 / t stays at 1 / far)
  const bool duck_possible = zc - K.duck_radius > near && zc - K.duck_radius < far && s_blocked == 0;
  float* img = out + (size_t)env * 2 * res * res;
  float* img1 = img + (size_t)res * res;
  const int tpr = (res + kRTileW - 1) / kRTileW;
  FW_RP(4);
  // (`wave` went through v_readfirstlane above: the strip counters, the row loop and their exits are scalar control flow -- with a
  // wave number the compiler takes for divergent, every loop here carried an exec-mask protocol and the strip index a vector division)
  // STAGE: the image leaves through LDS (stage_px pixels per channel behind the coordinate table; a build of its own so that the
  // direct form keeps its 64 registers): a strip is four
  // pixels wide, so a wave's store instruction wrote 16 rows x 16 bytes -- sixteen partial lines per instruction, and the stores alone
  // (no pixel arithmetic) took 12 of this kernel's 25.6 us at 4096 x 32 x 32.  A band of rows that fits the stage is rendered into LDS,
  // and the workgroup writes it out as whole rows, 16 bytes per lane and 1 KB per wave instruction.
  float* s_img = reinterpret_cast<float*>(smem_raw + (size_t)((res + 1) & ~1) * sizeof(double));
  const int band = STAGE ? min(kRTileH, (stage_px / res) & ~15) : kRTileH;       // rows per pass (a multiple of the 16 rows one instruction covers)
  for (int y0 = 0; y0 < res; y0 += band) {
  const int yend = min(y0 + band, res);
  for (int tx = wave; tx < tpr; tx += nwaves) {
    const int x0 = tx * kRTileW, x1 = min(x0 + kRTileW, res) - 1, y1 = yend - 1;
    // ---- cull: lane c looks at cylinder c -- a half-plane that holds none of the four corners of the tile holds no pixel of it
    // (the largest corner value of an affine form is its constant plus the larger end of either coordinate's term) ----
    unsigned int alive = 0u;
    const float a0 = (float)s_ab[x0], a1 = (float)s_ab[x1], b0 = (float)s_ab[y0], b1 = (float)s_ab[y1];
    // (wave-uniform: a strip the silhouette's box does not reach skips the silhouette test of its pixels)
    const bool duck_tile = duck_possible && a0 <= s_dbox[1] && a1 >= s_dbox[0] && b0 <= s_dbox[3] && b1 >= s_dbox[2];
    {
      bool keep = false;
      if (lane < nob) {
        const float m = -s_margin[lane];
        bool in[2];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          const float g0 = s_wedge[lane][hp][0], g1 = s_wedge[lane][hp][1], g2 = s_wedge[lane][hp][2];
          in[hp] = g0 + fmaxf(a0 * g1, a1 * g1) + fmaxf(b0 * g2, b1 * g2) >= m;
        }
        keep = in[0] && in[1];
      }
      alive = (unsigned int)__ballot(keep);          // (FW_MAX_OBSTACLES <= 32 cylinders: the low word)
    }
    // ---- pixels of the tile ----
    const int xi = x0 + (lane & (kRTileW - 1));
    const int krows = (yend - y0 + 64 / kRTileW - 1) / (64 / kRTileW);
#pragma unroll 1
    for (int k = 0; k < krows; ++k) {
      const int yi = y0 + (lane >> 2) + (64 / kRTileW) * k;
      if (xi < res && yi < yend) {
      const double a = s_ab[xi], b = s_ab[yi];
      bool is_duck = false;
      double inv_t = 0.0;                                                            // 1 / (view-axis depth) of the nearest fragment
      if (duck_tile) {
        const double q = 1.0 + a * a + b * b, p = zc + a * xc + b * yc, disc = p * p - q * k2;       // (the checker's expressions: the silhouette)
        if (disc >= 0.0 && p > 0.0) {
          const double num = p - M<double>::sqrt_(disc);
          if (num > 0.0) {
            inv_t = M<double>::div_(q, num);
            is_duck = inv_t < inv_near && inv_t > inv_far;
            if (::fabs(inv_t - inv_near) <= 1e-9 * inv_near || ::fabs(inv_t - inv_far) <= 1e-9 * inv_far) {       // at a clip plane: the checker's own sequence decides
              const double t_duck = (p - ::sqrt(disc)) / q;
              is_duck = t_duck > near && t_duck < far;
            }
          }
        }
      }
      if (!is_duck) {
        const double dwx = fma(b, Hd[0], fma(a, Hr[0], Hf[0])), dwy = fma(b, Hd[1], fma(a, Hr[1], Hf[1])), dwz = fma(b, Hd[2], fma(a, Hr[2], Hf[2]));
        // ray_depth(): best = far, or the ground at t = -cam z / dw z (dw z < 0, t > 0): kground <= 0, so 1 / t = dw z kground is positive
        // exactly where dw z < 0, and one maximum says both
        inv_t = ::fmax(inv_far, dwz * kground);
        const double qa = fma(dwx, dwx, dwy * dwy);
        if (qa > 0.0) {
          unsigned int m = alive;
          while (m) {                                                                // ... the cylinders this tile can see
            const int o = __ffs((int)m) - 1;
            m &= m - 1u;
            const double ox = s_cyl[o][0], oy = s_cyl[o][1], cc = s_cyl[o][2], hh = s_cyl[o][3];
            const double hb = fma(ox, dwx, oy * dwy);                                // t = (-hb - sqrt(hb^2 - qa cc)) / qa
            const double disc = fma(hb, hb, -(qa * cc));
            if (disc < 0.0) continue;
            const double num = -hb - M<double>::sqrt_(disc);
            if (!(num > 0.0)) continue;
            const double ic = M<double>::div_(qa, num);
            const double zi = fma(cam[2], ic, dwz);                                  // z / t of the hit: 0 <= z <= height
            const double it2 = ::fmax(inv_t, ic);
            inv_t = (zi >= 0.0 && zi <= hh * ic) ? it2 : inv_t;
          }
        }
      }
      inv_t = ::fmax(::fmin(inv_t, inv_near), inv_far);                              // t clipped to [near, far]
      const double dv = db_c1 * fma(-near, inv_t, 1.0);                              // far (t - near) / (t (far - near))
      if (STAGE) {
        const int sp = (yi - y0) * res + xi;
        s_img[sp] = is_duck ? 1.0f : 0.0f;
        s_img[stage_px + sp] = (float)dv;
      } else {
        const unsigned int off = (unsigned int)(yi * res + xi) * 4u;                // (res <= 1024: an unsigned 32-bit byte offset on a scalar base)
        *reinterpret_cast<float*>(reinterpret_cast<char*>(img) + off) = is_duck ? 1.0f : 0.0f;
        *reinterpret_cast<float*>(reinterpret_cast<char*>(img1) + off) = (float)dv;
      }
      }
    }
  }
  if (STAGE) {
    __syncthreads();
    const int nfl = (yend - y0) * res;                                              // the band's rows are one run of memory per channel
    float* g0 = img + (size_t)y0 * res;
    float* g1 = img1 + (size_t)y0 * res;
    if ((res & 3) == 0) {                                                           // (16-byte pieces: every run starts on a multiple of 4 pixels)
      for (int i = t * 4; i < nfl; i += nwaves * 256) {
        *reinterpret_cast<float4*>(g0 + i) = *reinterpret_cast<const float4*>(s_img + i);
        *reinterpret_cast<float4*>(g1 + i) = *reinterpret_cast<const float4*>(s_img + stage_px + i);
      }
    } else {
      for (int i = t; i < nfl; i += nwaves * 64) { g0[i] = s_img[i]; g1[i] = s_img[stage_px + i]; }
    }
    if (yend < res) __syncthreads();                                                // (the next band overwrites the stage)
  }
  }
  FW_RP(5);
}

}  // namespace fwsim
