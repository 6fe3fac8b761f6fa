// fw_render: the FPV image of the analytic scene (sphere duck, cylinder obstacles, ground plane, sky) that a14's image
// functionals are computed on, written out pixel by pixel for a network to consume -- what Camera.capture_image() hands the
// env in the reference (segImg / depthImg, envs/fixedwing_objlock_env.py:603-622) and what its CNN path feeds a detector
// (envs/fixedwing_envs/objlock_yolo_env.py:646-716: a network's mask replaces segImg).
//   out[env][0][y][x] = 1.0 where the pixel's ray hits the duck sphere between the clip planes (none if a cylinder blocks the
//                       line of sight to its centre: the scene's binary occlusion rule), else 0.0
//   out[env][1][y][x] = depth-buffer value in [0, 1] of the nearest fragment (duck pixels: the sphere; others: ground or
//                       cylinder; sky = 1.0), far (t - near) / (t (far - near)) with t clipped to [near, far]
// at `res` x `res` pixels of the same body-fixed camera (FOV, tilt, offset of fw_config; the focal length scales with the
// width), for the env's CURRENT pose.  One workgroup per env: the pose, the duck's camera-frame centre and the occlusion flag are
// computed once into LDS next to the cylinder table, then the 256 threads walk the pixels (x fastest: coalesced float stores).
// The arithmetic is double whatever the handle's dtype and is written statement by statement like the CPU checker's render (IEEE sqrt and
// division, no FMA contraction): the duck mask is an exact comparison against 0 at the silhouette, so the test asks for the
// same bits, not for a tolerance.
#pragma once
#include "fwsim_device.hpp"
#include "fwsim_objlock.hpp"

namespace fwsim {

struct RenderC {            // camera constants in double (built on the host from fw_config)
  double cam_f[3], cam_r[3], cam_d[3], cam_off[3];
  double tan_half_fov, near_, far_, duck_radius, obst_radius;
};

template <typename T>
__global__ __launch_bounds__(256) void fw_render_kernel(const T* __restrict__ r, int tile, int n_envs, RenderC K, int res,
                                                        float* __restrict__ out) {
#pragma clang fp contract(off)                    // this kernel only (block scope): multiply-adds stay two roundings, as in the CPU checker's C
  __shared__ double s_c[24];                      // R[9] cam[3] zc xc yc k2 | flag nob
  __shared__ double s_cyl[FW_MAX_OBSTACLES][3];
  const int env = blockIdx.x, t = threadIdx.x;
  if (env >= n_envs) return;
  auto fld = [&](int f) { return (double)r[tile_index(tile, RF_COUNT, f, env)]; };
  int nob = (int)fld(RF_TASK + FW_ST_NUM_OBST);
  nob = nob < 0 ? 0 : (nob > FW_MAX_OBSTACLES ? FW_MAX_OBSTACLES : nob);
  if (t < 3 * FW_MAX_OBSTACLES) s_cyl[t / 3][t % 3] = (t / 3 < nob) ? fld(RF_TASK + FW_ST_OBST + t) : 0.0;
  __syncthreads();
  if (t == 0) {
    const double x = fld(RF_QUAT), y = fld(RF_QUAT + 1), z = fld(RF_QUAT + 2), w = fld(RF_QUAT + 3);
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;                 // btMatrix3x3::setRotation
    const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs;
    const double xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    double R[9] = { 1.0 - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0 - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0 - (xx + yy) };
    double cam[3];
    for (int k = 0; k < 3; ++k) cam[k] = fld(RF_POS + k) + (R[3 * k] * K.cam_off[0] + R[3 * k + 1] * K.cam_off[1] + R[3 * k + 2] * K.cam_off[2]);
    const double Rd = K.duck_radius;
    const double C[3] = { fld(RF_TASK + FW_ST_DUCK_POS), fld(RF_TASK + FW_ST_DUCK_POS + 1), fld(RF_TASK + FW_ST_DUCK_POS + 2) + Rd };
    const double relw[3] = { C[0] - cam[0], C[1] - cam[1], C[2] - cam[2] };
    double relb[3];
    for (int k = 0; k < 3; ++k) relb[k] = R[k] * relw[0] + R[3 + k] * relw[1] + R[6 + k] * relw[2];       // R^T relw
    const double zc = relb[0] * K.cam_f[0] + relb[1] * K.cam_f[1] + relb[2] * K.cam_f[2];
    const double xc = relb[0] * K.cam_r[0] + relb[1] * K.cam_r[1] + relb[2] * K.cam_r[2];
    const double yc = relb[0] * K.cam_d[0] + relb[1] * K.cam_d[1] + relb[2] * K.cam_d[2];
    const double k2 = zc * zc + xc * xc + yc * yc - Rd * Rd;
    bool blocked = false;                                                          // occluded(): the segment cam -> C
    for (int o = 0; o < nob; ++o) {
      const double ox = cam[0] - s_cyl[o][0], oy = cam[1] - s_cyl[o][1], hh = s_cyl[o][2];
      const double a = relw[0] * relw[0] + relw[1] * relw[1], b = 2.0 * (ox * relw[0] + oy * relw[1]);
      const double cc = ox * ox + oy * oy - K.obst_radius * K.obst_radius;
      if (a <= 0.0) continue;
      const double disc = b * b - 4.0 * a * cc;
      if (disc < 0.0) continue;
      const double tt = (-b - ::sqrt(disc)) / (2.0 * a);
      if (tt <= 0.0 || tt >= 1.0) continue;
      const double zz2 = cam[2] + tt * relw[2];
      if (zz2 >= 0.0 && zz2 <= hh) blocked = true;
    }
    for (int k = 0; k < 9; ++k) s_c[k] = R[k];
    for (int k = 0; k < 3; ++k) s_c[9 + k] = cam[k];
    s_c[12] = zc; s_c[13] = xc; s_c[14] = yc; s_c[15] = k2;
    s_c[16] = (zc - Rd > K.near_ && zc - Rd < K.far_ && !blocked) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double W = (double)res, F = 0.5 * W / K.tan_half_fov, u0 = 0.5 * (W - 1.0), near = K.near_, far = K.far_;
  const double zc = s_c[12], xc = s_c[13], yc = s_c[14], k2 = s_c[15];
  const bool duck_possible = s_c[16] != 0.0;
  const double cam0 = s_c[9], cam1 = s_c[10], cam2 = s_c[11];
  float* img = out + (size_t)env * 2 * res * res;
  auto depth_buffer_of = [&](double tv) {
#pragma clang fp contract(off)
    if (tv < near) tv = near;
    if (tv > far) tv = far;
    return far * (tv - near) / (tv * (far - near));
  };
  for (int px = t; px < res * res; px += 256) {
    const int yi = px / res, xi = px - yi * res;
    const double a = ((double)xi - u0) / F, b = ((double)yi - u0) / F;
    bool is_duck = false;
    double t_duck = 0.0;
    if (duck_possible) {
      const double q = 1.0 + a * a + b * b, p = zc + a * xc + b * yc, disc = p * p - q * k2;
      if (disc >= 0.0 && p > 0.0) { t_duck = (p - ::sqrt(disc)) / q; is_duck = t_duck > near && t_duck < far; }
    }
    double dv;
    if (is_duck) dv = depth_buffer_of(t_duck);
    else {
      double db[3], dw[3];
      for (int k = 0; k < 3; ++k) db[k] = K.cam_f[k] + a * K.cam_r[k] + b * K.cam_d[k];
      for (int k = 0; k < 3; ++k) dw[k] = s_c[3 * k] * db[0] + s_c[3 * k + 1] * db[1] + s_c[3 * k + 2] * db[2];
      double best = far;                                                           // ray_depth()
      if (dw[2] < 0.0) { const double tg = -cam2 / dw[2]; if (tg > 0.0 && tg < best) best = tg; }
      for (int o = 0; o < nob; ++o) {
        const double ox = cam0 - s_cyl[o][0], oy = cam1 - s_cyl[o][1], hh = s_cyl[o][2];
        const double qa = dw[0] * dw[0] + dw[1] * dw[1], qb = 2.0 * (ox * dw[0] + oy * dw[1]);
        const double cc = ox * ox + oy * oy - K.obst_radius * K.obst_radius;
        if (qa <= 0.0) continue;
        const double disc = qb * qb - 4.0 * qa * cc;
        if (disc < 0.0) continue;
        const double tc = (-qb - ::sqrt(disc)) / (2.0 * qa);
        if (tc <= 0.0) continue;
        const double zz2 = cam2 + tc * dw[2];
        if (zz2 < 0.0 || zz2 > hh) continue;
        if (tc < best) best = tc;
      }
      dv = depth_buffer_of(best < near ? near : best);
    }
    img[px] = is_duck ? 1.0f : 0.0f;
    img[(size_t)res * res + px] = (float)dv;
  }
}

}  // namespace fwsim
