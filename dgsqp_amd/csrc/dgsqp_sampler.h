// Rejection samplers of the Monte-Carlo scripts on the device: random placement of the cars, PID
// warm start, collision check along the warm start, compaction of the accepted candidates in candidate order.
//   first_segment  scripts/DGSQP_ALGAMES_monte_carlo_chicane.py:384-404 (curve.py the same): car 1 on the first track segment, car 2 at
//                  1.2 obstacle distances in a random direction
//   independent    scripts/DGSQP_monte_carlo_agents.py:262-308: every car independently on the first segment
//   circuit        scripts/DGSQP_comp_monte_carlo.py:365-382: car 1 anywhere on the circuit, the others within 1.2 obstacle distances along it
//   merge          scripts/DGSQP_merge_monte_carlo.py:429-473: cars around their nominal places on the lane / the ramp, zero inputs
// The random numbers come from a COUNTER-BASED generator (Philox4x32-10, Salmon et al. 2011): uniform k of candidate c is a pure
// function of (seed, c, k), so candidates can be drawn in any order and in any batch size, and the numpy mirror
// (dgsqp_amd/sampler.py) reproduces the stream bit for bit.
#pragma once

struct DgPhilox { unsigned int v[4]; };
__host__ __device__ inline void dg_philox_round(unsigned int* c, unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0, hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
  const unsigned int n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__host__ __device__ inline DgPhilox dg_philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0, unsigned int k1) {
  DgPhilox r{{c0, c1, c2, c3}};
  for (int i = 0; i < 10; i++) {
    dg_philox_round(r.v, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return r;
}
// uniform k (0-based) of candidate c: 53 random bits, the convention of numpy's random_sample ((a >> 5) 2^26 + (b >> 6)) / 2^53
__host__ __device__ inline double dg_uniform(unsigned long long seed, unsigned long long cand, int k) {
  const DgPhilox r = dg_philox4x32_10((unsigned int)cand, (unsigned int)(cand >> 32), (unsigned int)(k >> 1), 0u, (unsigned int)seed, (unsigned int)(seed >> 32));
  const unsigned int a = r.v[(k & 1) * 2], b = r.v[(k & 1) * 2 + 1];
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// Frenet -> global on an arc track (radius_arclength_track.py:752-807), key points (x, y, psi, cumulative length, segment length,
// signed curvature) as the reference keeps them
__device__ inline double dg_wrap_pi(double a) {
  const double twopi = 6.283185307179586;
  a = a - twopi * floor((a + 3.141592653589793) / twopi);
  return a;
}
__device__ inline void dg_local_to_global(const dgsqp_sampler_t& S, double s, double ey, double* x, double* y) {
  const double L = S.key_pts[S.n_key - 1][3];
  while (s < 0) s += L;
  while (s >= L) s -= L;
  int i0 = 0;
  for (int i = 0; i < S.n_key - 1; i++) if (s >= S.key_pts[i][3]) i0 = i;
  const int i1 = i0 + 1;
  const double xs = S.key_pts[i0][0], ys = S.key_pts[i0][1], psis = S.key_pts[i0][2];
  const double xf = S.key_pts[i1][0], yf = S.key_pts[i1][1], psif = S.key_pts[i1][2], curv = S.key_pts[i1][5], seg = S.key_pts[i1][4];
  const double d = s - S.key_pts[i0][3];
  const double hp = 1.5707963267948966;
  if (curv == 0.0) {
    *x = xs + (xf - xs) * d / seg + ey * cos(psif + hp);
    *y = ys + (yf - ys) * d / seg + ey * sin(psif + hp);
  } else {
    const double r = 1.0 / curv, sgn = r >= 0 ? 1.0 : -1.0, ar = fabs(r);
    const double xc = xs + ar * cos(psis + sgn * hp), yc = ys + ar * sin(psis + sgn * hp);
    const double span = d / ar;
    const double an = dg_wrap_pi(psis + sgn * hp);
    const double ang = -(an >= 0 ? 1.0 : -1.0) * (3.141592653589793 - fabs(an));
    *x = xc + (ar - sgn * ey) * cos(ang + sgn * span);
    *y = yc + (ar - sgn * ey) * sin(ang + sgn * span);
  }
}

// one lane per candidate: placement -> q0[cand][n_q] (+ ok[cand] = the placement itself is admissible)
__global__ void dg_sample_place_kernel(int64_t n, unsigned long long c0, dgsqp_sampler_t S, double* __restrict__ q0, int32_t* __restrict__ ok) {
  const DgProb& D = dg_prob;
  const double pi = 3.141592653589793;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long c = c0 + (unsigned long long)i;
    double* q = q0 + i * D.nq;
    for (int t = 0; t < D.nq; t++) q[t] = 0.0;
    int good = 1;
    auto U = [&](int k) { return dg_uniform(S.seed, c, k); };
    auto put = [&](int a, double s, double ey, double v, double epsi) {   // Frenet-frame models: [x, y, v, (..), e_psi, s, e_y]
      double x, y;
      dg_local_to_global(S, s, ey, &x, &y);
      double* qa = q + D.qoff[a];
      const int nqa = D.nqa[a];
      qa[0] = x; qa[1] = y; qa[2] = v; qa[nqa == 8 ? 5 : 3] = epsi; qa[nqa - 2] = s; qa[nqa - 1] = ey;
    };
    if (S.kind == DGSQP_SAMPLER_FIRST_SEGMENT) {
      const double s1 = fmax(0.1, U(0) * S.seg0_len), ey1 = U(1) * S.half_width * 2 - S.half_width, v1 = U(2) + 2;
      const double dd = 2 * pi * U(3);
      const double s2 = s1 + 1.2 * S.obs_d * cos(dd), ey2 = ey1 + 1.2 * S.obs_d * sin(dd), v2 = U(4) + 2;
      good = (s2 >= 0) && (fabs(ey2) <= S.half_width);
      put(0, s1, ey1, v1, 0.0);
      put(1, good ? s2 : s1, good ? ey2 : ey1, v2, 0.0);          // (a rejected placement is never used: keep its warm start well defined)
    } else if (S.kind == DGSQP_SAMPLER_INDEPENDENT) {
      for (int a = 0; a < D.M; a++) put(a, fmax(0.1, U(3 * a) * S.seg0_len), U(3 * a + 1) * S.half_width * 2 - S.half_width, U(3 * a + 2) + 2, 0.0);
    } else if (S.kind == DGSQP_SAMPLER_CIRCUIT) {
      const double s1 = S.key_pts[S.n_key - 1][3] * U(0), v1 = 2.0 + (U(2) - 0.5);
      put(0, s1, S.half_width * (2 * U(1) - 1), v1, 5.0 * (2 * U(3) - 1) * pi / 180);
      for (int a = 1; a < D.M; a++)
        put(a, s1 + 1.2 * S.obs_d * (2 * U(4 * a) - 1), S.half_width * (2 * U(4 * a + 1) - 1), (1 + 0.25 * (2 * U(4 * a + 2) - 1)) * v1, 5.0 * (2 * U(4 * a + 3) - 1) * pi / 180);
    } else {       // merge: unicycles [x, y, v, psi]
      const double mw = 0.3, mp = 1.5, th = pi / 12;
      const double x5 = mp, x7 = mp + mw / sin(th);
      for (int a = 0; a < D.M; a++) {
        double* qa = q + D.qoff[a];
        const double xn = S.x_nom[a];
        if (a % 3 != 2) {
          qa[0] = xn + 0.5 * U(4 * a) - 0.25; qa[1] = 0.15 + 0.1 * U(4 * a + 1) - 0.05;
          qa[2] = 0.3 * (1 + 0.06 * U(4 * a + 2) - 0.03); qa[3] = (5 * U(4 * a + 3) - 2.5) * pi / 180;
        } else {
          const double yn = -((x7 + x5) / 2 - xn) * tan(th);
          const double sr = 0.5 * U(4 * a) - 0.25, er = 0.1 * U(4 * a + 1) - 0.05;
          qa[0] = xn + sr * cos(th) - er * sin(th); qa[1] = yn + sr * sin(th) + er * cos(th);
          qa[2] = 0.3 * (1 + 0.06 * U(4 * a + 2) - 0.03); qa[3] = pi / 12 + (5 * U(4 * a + 3) - 2.5) * pi / 180;
        }
      }
    }
    ok[i] = good;
  }
}

// merge: zero-input trajectories by the joint model's own integrator (q_ws [n][(N+1) n_q]).  For the script's three cars its quirk is
// reproduced: car 3's check trajectory stays at the origin (merge.py:471-472 assigns car2_q_ws[0] twice)
__global__ void dg_sample_zero_rollout_kernel(int64_t n, const double* __restrict__ q0, double* __restrict__ q_ws) {
  const DgProb& D = dg_prob;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n * D.M; it += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = it / D.M;
    const int a = (int)(it % D.M);
    const int nqa = D.nqa[a], qo = D.qoff[a];
    double* out = q_ws + b * (int64_t)(D.N + 1) * D.nq + qo;
    Ty<0> q[4], u[2], qn[4];
    const bool origin = D.M == 3 && a == 2;
    for (int i = 0; i < 4; i++) q[i].c[0] = (origin || i >= nqa) ? 0.0 : q0[b * D.nq + qo + i];
    u[0].c[0] = 0.0; u[1].c[0] = 0.0;
    for (int i = 0; i < nqa && i < 4; i++) out[i] = q[i].c[0];
    for (int k = 0; k < D.N; k++) {
      dev_fd<0, 4>(D.P, D.P.agents[a], q, u, qn);
      for (int i = 0; i < 4; i++) q[i] = qn[i];
      for (int i = 0; i < nqa && i < 4; i++) out[(int64_t)(k + 1) * D.nq + i] = q[i].c[0];
    }
  }
}

// accepted = placement ok and no collision: exclusive prefix count in candidate order (one workgroup, n <= a few 10^4), then the
// accepted candidates are copied behind the `have` scenarios already collected -- candidate order, like the sequential scripts
__global__ void __launch_bounds__(1024)
dg_sample_scan_kernel(int64_t n, const int32_t* __restrict__ ok, const int32_t* __restrict__ collide, int32_t* __restrict__ pos, int32_t* __restrict__ count) {
  __shared__ int part[1024];
  const int t = threadIdx.x, nt = blockDim.x;
  const int64_t per = (n + nt - 1) / nt, lo = t * per, hi = lo + per < n ? lo + per : n;
  int s = 0;
  for (int64_t i = lo; i < hi; i++) s += (ok[i] && !collide[i]) ? 1 : 0;
  part[t] = s;
  __syncthreads();
  if (t == 0) { int acc = 0; for (int i = 0; i < nt; i++) { const int v = part[i]; part[i] = acc; acc += v; } *count = acc; }
  __syncthreads();
  int p = part[t];
  for (int64_t i = lo; i < hi; i++) { const bool acc = ok[i] && !collide[i]; pos[i] = acc ? p : -1; p += acc ? 1 : 0; }
}
__global__ void dg_sample_gather_kernel(int64_t n, int64_t have, int64_t B, const int32_t* __restrict__ pos, const double* __restrict__ q0,
                                        const double* __restrict__ u_cand, double* __restrict__ x0_out, double* __restrict__ u_out) {
  const DgProb& D = dg_prob;
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int p = pos[i];
    if (p < 0 || have + p >= B) continue;
    const int64_t o = have + p;
    for (int k = threadIdx.x; k < D.nq; k += blockDim.x) x0_out[o * D.nq + k] = q0[i * D.nq + k];
    for (int k = threadIdx.x; k < D.n; k += blockDim.x) u_out[o * D.n + k] = u_cand ? u_cand[i * D.n + k] : 0.0;
  }
}
