// fw_render: the FPV image of the analytic scene (sphere duck, cylinder obstacles, ground plane, sky) that a14's image
// functionals are computed on, written out pixel by pixel for a network to consume -- what Camera.capture_image() hands the
// env in the reference (segImg / depthImg, envs/fixedwing_objlock_env.py:603-622) and what its CNN path feeds a detector
// (envs/fixedwing_envs/objlock_yolo_env.py:646-716: a network's mask replaces segImg).
//   out[env][0][y][x] = 1.0 where the pixel's ray hits the duck sphere between the clip planes (none if a cylinder blocks the
//                       line of sight to its centre: the scene's binary occlusion rule), else 0.0
//   out[env][1][y][x] = depth-buffer value in [0, 1] of the nearest fragment (duck pixels: the sphere; others: ground or
//                       cylinder; sky = 1.0), far (t - near) / (t (far - near)) with t clipped to [near, far]
// at `res` x `res` pixels of the same body-fixed camera (FOV, tilt, offset of fw_config; the focal length scales with the
// width), for the env's CURRENT pose.  One workgroup (four waves) per env; a wave takes 16 x 16-pixel tiles.
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
//   * every thread derives the pose / duck constants for itself (same loads, same arithmetic: no broadcast barrier for them).
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

template <typename T>
__global__ __launch_bounds__(256) void fw_render_kernel(const T* __restrict__ r, int tile, int n_envs, RenderC K, int res,
                                                        float* __restrict__ out) {
#pragma clang fp contract(off)                    // this kernel only: multiply-adds stay two roundings, as in the CPU checker's C
  __shared__ double s_pose[28];                   // R[9] cam[3] zc xc yc k2 | Hf Hr Hd (the camera axes in the world frame)
  __shared__ double s_cyl[FW_MAX_OBSTACLES][4];   // ox, oy, cc, height: cam - axis (horizontal), |.|^2 - radius^2
  __shared__ float s_wedge[FW_MAX_OBSTACLES][2][3];   // the two tangent half-planes as affine forms g(a, b) = g0 + a g1 + b g2 (>= 0 inside); [.][0][0] = +inf: never cull
  __shared__ float s_margin[FW_MAX_OBSTACLES];
  __shared__ int s_blocked;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* s_ab = reinterpret_cast<double*>(smem_raw);      // [res]: (i - u0) / F, the image-plane coordinate of pixel column / row i (the image is square)
  const int env = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (env >= n_envs) return;
  auto fld = [&](int f) { return (double)r[tile_index(tile, RF_COUNT, f, env)]; };
  int nob = (int)fld(RF_TASK + FW_ST_NUM_OBST);
  nob = nob < 0 ? 0 : (nob > FW_MAX_OBSTACLES ? FW_MAX_OBSTACLES : nob);
  const double W = (double)res, F = 0.5 * W / K.tan_half_fov, u0 = 0.5 * (W - 1.0), near = K.near_, far = K.far_;
  for (int i = t; i < res; i += (int)blockDim.x) s_ab[i] = ((double)i - u0) / F;      // (one IEEE division per column, not four per pixel)
  if (wave == (env & ((int)(blockDim.x >> 6) - 1))) {
    // the pose, the duck in the camera frame, the ray basis: every lane of ONE wave for itself (no broadcast inside the wave), lane 0
    // leaves them for the other waves.  Which wave rotates with the env: a workgroup's wave i runs on SIMD i, and with the pixel loop
    // as short as it now is the set-up would otherwise queue sixteen deep on SIMD 0 of every CU
    double R[9], cam[3], relw[3], zc, xc, yc, k2, Hf[3], Hr[3], Hd[3];
    const double x = fld(RF_QUAT), y = fld(RF_QUAT + 1), z = fld(RF_QUAT + 2), w = fld(RF_QUAT + 3);
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;                 // btMatrix3x3::setRotation
    const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs;
    const double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1.0 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy; R[3] = xy + wz; R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1.0 - (xx + yy);
    for (int k = 0; k < 3; ++k) cam[k] = fld(RF_POS + k) + (R[3 * k] * K.cam_off[0] + R[3 * k + 1] * K.cam_off[1] + R[3 * k + 2] * K.cam_off[2]);
    const double Rd = K.duck_radius;
    const double C[3] = { fld(RF_TASK + FW_ST_DUCK_POS), fld(RF_TASK + FW_ST_DUCK_POS + 1), fld(RF_TASK + FW_ST_DUCK_POS + 2) + Rd };
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
    if (lane == 0) {
      for (int k = 0; k < 9; ++k) s_pose[k] = R[k];
      for (int k = 0; k < 3; ++k) s_pose[9 + k] = cam[k];
      s_pose[12] = zc; s_pose[13] = xc; s_pose[14] = yc; s_pose[15] = k2;
      for (int k = 0; k < 3; ++k) { s_pose[16 + k] = Hf[k]; s_pose[19 + k] = Hr[k]; s_pose[22 + k] = Hd[k]; }
      s_pose[25] = cam[2] > 0.0 ? -1.0 / cam[2] : 0.0;
    }
    bool blocked_any = false;
    if (lane < FW_MAX_OBSTACLES) {                   // lane o: cylinder o
      bool blocked = false;
      double ox = 0.0, oy = 0.0, cc = 1.0, hh = 0.0;
      if (lane < nob) {
        const double ax = fld(RF_TASK + FW_ST_OBST + 3 * lane), ay = fld(RF_TASK + FW_ST_OBST + 3 * lane + 1);
        hh = fld(RF_TASK + FW_ST_OBST + 3 * lane + 2);
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
        const float ex = (float)-ox, ey = (float)-oy, tb = (float)(K.obst_radius / ::sqrt(cc));
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
    if (lane == 0) s_blocked = blocked_any ? 1 : 0;
  }
  __syncthreads();
  double cam[3], Hf[3], Hr[3], Hd[3];
  for (int k = 0; k < 3; ++k) { cam[k] = s_pose[9 + k]; Hf[k] = s_pose[16 + k]; Hr[k] = s_pose[19 + k]; Hd[k] = s_pose[22 + k]; }
  const double zc = s_pose[12], xc = s_pose[13], yc = s_pose[14], k2 = s_pose[15];
  const double inv_near = K.inv_near, inv_far = K.inv_far, db_c1 = K.db_c1;
  const double kground = s_pose[25];                                                 // -1 / (camera height); 0 for a camera at or below the ground (it sees none: 1 / t stays at 1 / far)
  const bool duck_possible = zc - K.duck_radius > near && zc - K.duck_radius < far && s_blocked == 0;
  float* img = out + (size_t)env * 2 * res * res;
  const int tpr = (res + kRTileW - 1) / kRTileW, ntiles = tpr * ((res + kRTileH - 1) / kRTileH);
  for (int tl = wave; tl < ntiles; tl += (int)(blockDim.x >> 6)) {
    const int ty = tl / tpr, tx = tl - ty * tpr;
    const int x0 = tx * kRTileW, y0 = ty * kRTileH, x1 = min(x0 + kRTileW, res) - 1, y1 = min(y0 + kRTileH, res) - 1;
    // ---- cull: lane c looks at cylinder c -- a half-plane that holds none of the four corners of the tile holds no pixel of it ----
    unsigned int alive = 0u;
    {
      bool keep = false;
      if (lane < nob) {
        const float a0 = (float)s_ab[x0], a1 = (float)s_ab[x1], b0 = (float)s_ab[y0], b1 = (float)s_ab[y1];
        const float m = -s_margin[lane];
        bool in[2];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          const float g0 = s_wedge[lane][hp][0], g1 = s_wedge[lane][hp][1], g2 = s_wedge[lane][hp][2];
          const float u0_ = g0 + a0 * g1, u1_ = g0 + a1 * g1, v0_ = b0 * g2, v1_ = b1 * g2;
          in[hp] = u0_ + v0_ >= m || u1_ + v0_ >= m || u0_ + v1_ >= m || u1_ + v1_ >= m;
        }
        keep = in[0] && in[1];
      }
      alive = (unsigned int)__ballot(keep);          // (FW_MAX_OBSTACLES <= 32 cylinders: the low word)
    }
    // ---- pixels of the tile ----
    const int xi = x0 + (lane & (kRTileW - 1));
#pragma unroll 1
    for (int k = 0; k < kRTileW * kRTileH / 64; ++k) {
      const int yi = y0 + (lane >> 2) + (64 / kRTileW) * k;
      if (y0 + (64 / kRTileW) * k >= res) break;                                       // (wave-uniform: the strip is taller than the image)
      if (xi >= res || yi >= res) continue;
      const double a = s_ab[xi], b = s_ab[yi];
      bool is_duck = false;
      double inv_t = 0.0;                                                            // 1 / (view-axis depth) of the nearest fragment
      if (duck_possible) {
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
        inv_t = inv_far;                                                             // ray_depth(): best = far ...
        const double ig = dwz * kground;                                             // ... the ground at t = -cam z / dw z (dw z < 0, t > 0) ...
        if (dwz < 0.0 && ig > inv_t) inv_t = ig;
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
            if (zi < 0.0 || zi > hh * ic) continue;
            if (ic > inv_t) inv_t = ic;
          }
        }
      }
      inv_t = inv_t > inv_near ? inv_near : inv_t;                                   // t clipped to [near, far]
      inv_t = inv_t < inv_far ? inv_far : inv_t;
      const double dv = db_c1 * fma(-near, inv_t, 1.0);                              // far (t - near) / (t (far - near))
      const int px = yi * res + xi;
      img[px] = is_duck ? 1.0f : 0.0f;
      img[(size_t)res * res + px] = (float)dv;
    }
  }
}

}  // namespace fwsim
