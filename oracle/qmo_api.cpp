// qmo_api.cpp -- TEST INFRASTRUCTURE ONLY.  extern "C" entry points of the CPU oracle for ctypes (tests/, smoke(),
// bench.py cpu_baseline).  PARITY UNPINNED: see qmo_core.h.  The product (qm_door_amd/) never links or loads this.
#include <malloc.h>

#include <chrono>
#include <cstdio>
#include <memory>
#include <thread>

#include "qmo_mpc.h"
#include "qmo_wbc.h"

using namespace qmo;

// force tracking (own formulation): contact reference [K][6] attached to every Target / end-effector force handed to every WBC update
// built by the entry points below until cleared (null).  Set from the tests' thread before a call; not part of the timed baselines.
static const double* g_contactRef = nullptr;
static const double* g_wbcEeForce = nullptr;
// solver state carried from WBC tick to WBC tick (the counterpart of qmgpu_wbc_args::working_set): [B][QMGPU_WBC_STATE_WORDS] words, instance i of the batch entry points
// (and the one instance of qmo_wbc_update) uses its own block; null: every tick cold.  Set from the tests' thread before a call.
static uint64_t* g_wbcWorkingSet = nullptr;

extern "C" {

void qmo_set_ee_contact_ref(const double* ref /* [K][6] or null */) { g_contactRef = ref; }
void qmo_set_wbc_ee_force(const double* f3 /* [3] or null */) { g_wbcEeForce = f3; }
void qmo_set_wbc_working_set(uint64_t* ws /* [B][QMGPU_WBC_STATE_WORDS] or null */) { g_wbcWorkingSet = ws; }

void qmo_flow_map(const qmgpu_problem* P, const double* x, const double* u, double* f) { flowMap<double>(P->model, P->settings.gravity, x, u, f); }

// max |difference| between the two derivative routes of the oracle (Dual<60> through everything vs 21 dual directions + closed-form
// linear columns) over the flow map and every auxiliary quantity the LQ approximation differentiates
double qmo_structured_vs_dual60(const qmgpu_problem* P, const double* x, const double* u) {
  static thread_local D60m xd[30], ud[30], fa[30], fb[30];
  for (int i = 0; i < 30; ++i) { xd[i] = D60m(x[i]); xd[i].d[i] = 1.0; ud[i] = D60m(u[i]); ud[i].d[30 + i] = 1.0; }
  FlowAux<D60m> aa, ab;
  flowMap<D60m>(P->model, P->settings.gravity, xd, ud, fa, &aa);
  flowMapStructured(P->model, P->settings.gravity, x, u, fb, &ab);
  double worst = 0.0;
  auto cmp = [&](const D60m& p, const D60m& q) { worst = std::max(worst, std::fabs(p.v - q.v)); for (int i = 0; i < 60; ++i) worst = std::max(worst, std::fabs(p.d[i] - q.d[i])); };
  for (int i = 0; i < 30; ++i) cmp(fa[i], fb[i]);
  for (int c = 0; c < 4; ++c) for (int a = 0; a < 3; ++a) { cmp(aa.footPos[c][a], ab.footPos[c][a]); cmp(aa.footVel[c][a], ab.footVel[c][a]); }
  for (int a = 0; a < 3; ++a) { cmp(aa.eePos[a], ab.eePos[a]); for (int b = 0; b < 3; ++b) cmp(aa.eeRot.m[a][b], ab.eeRot.m[a][b]); }
  return worst;
}
void qmo_set_flow_contact(double K, const double* env3) { g_eeContact = EeContact(); g_eeContact.K = K; if (env3) for (int a = 0; a < 3; ++a) g_eeContact.env[a] = env3[a]; }

// thread-seconds spent per phase (all threads) since the last call: LQ + projection, Riccati, line search, WBC model, WBC QPs
// (timing-grade build only; zeros in the checker build)
void qmo_time_split(double* out5) { std::lock_guard<std::mutex> lock(g_phaseMutex); for (int i = 0; i < PH_COUNT; ++i) { out5[i] = g_phaseSeconds[i]; g_phaseSeconds[i] = 0.0; } }
int qmo_is_fast_build(void) {
#ifdef QMO_FAST
  return 1;
#else
  return 0;
#endif
}

void qmo_flow_map_lin(const qmgpu_problem* P, const double* x, const double* u, double* f, double* A, double* B) {
  Mat Am, Bm;
  flowMapLinearization(*P, x, u, f, Am, Bm);
  Am.to(A); Bm.to(B);
}

void qmo_kinematics(const qmgpu_problem* P, const double* x, const double* u, double* footPos, double* footVel, double* eePos, double* eeQuat, double* com) {
  double f[30];
  FlowAux<double> aux;
  flowMap<double>(P->model, P->settings.gravity, x, u, f, &aux);
  for (int c = 0; c < 4; ++c) for (int a = 0; a < 3; ++a) { footPos[3 * c + a] = aux.footPos[c][a]; footVel[3 * c + a] = aux.footVel[c][a]; }
  for (int a = 0; a < 3; ++a) eePos[a] = aux.eePos[a];
  matrixToQuaternion(aux.eeRot, eeQuat);
  if (com) { Kin<double> k; forwardKinematics<double>(P->model, x + 6, k); for (int a = 0; a < 3; ++a) com[a] = k.comTotal[a]; }
}

void qmo_centroidal_matrix(const qmgpu_problem* P, const double* q, double* A /*6x24*/) {
  Kin<double> k; forwardKinematics<double>(P->model, q, k);
  double Am[6][NV]; centroidalMomentumMatrix(P->model, k, Am);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < NV; ++j) A[i * NV + j] = Am[i][j];
}

void qmo_input_weight(const qmgpu_problem* P, double* R) { inputWeight(*P).to(R); }

int qmo_mode_at(int nEv, const double* ev, const int32_t* modes, double t) { ModeSchedule ms{nEv, ev, modes}; return ms.modeAt(t); }
int qmo_node_mode_at(int nEv, const double* ev, const int32_t* modes, double t) { ModeSchedule ms{nEv, ev, modes}; return ms.nodeModeAt(t); }

void qmo_swing_reference(const qmgpu_problem* P, int nEv, const double* ev, const int32_t* modes, double t, double* zpos4, double* zvel4) {
  ModeSchedule ms{nEv, ev, modes};
  for (int c = 0; c < 4; ++c) swingReference(P->settings, ms, c, t, &zpos4[c], &zvel4[c]);
}

void qmo_reference_at(int K, const double* times, const double* states, double t, double* xref, double* eePos, double* eeQuat) {
  Target tg{K, times, states, nullptr};
  referenceAt(tg, t, xref, eePos, eeQuat);
}

// LQ approximation of one node before projection; matrices row-major 30x30, C/D nc x 30 packed (capacity 16 rows).
void qmo_lq_node(const qmgpu_problem* P, double t, double dt, const double* x, const double* u, const double* xnext, int terminal, int nEv, const double* ev,
                 const int32_t* modes, int K, const double* ttimes, const double* tstates, double* A, double* B, double* b, double* Q, double* R, double* q,
                 double* r, double* C, double* D, double* e, int32_t* nc, double* cost) {
  Problem pr{P, inputWeight(*P), ModeSchedule{nEv, ev, modes}, Target{K, ttimes, tstates, g_contactRef}};
  NodeLQ o;
  nodeLQ(pr, t, dt, x, u, xnext, terminal != 0, o);
  o.Q.to(Q);
  for (int i = 0; i < 30; ++i) q[i] = o.q[i];
  *cost = o.cost;
  *nc = o.nc;
  if (terminal) return;
  o.A.to(A); o.B.to(B); o.R.to(R);
  for (int i = 0; i < 30; ++i) { b[i] = o.b[i]; r[i] = o.r[i]; }
  if (o.nc > 0) { o.C.to(C); o.D.to(D); for (int i = 0; i < o.nc; ++i) e[i] = o.e[i]; }
}

// One SQP iteration. warmX/warmU may be NULL (initializer: x_k = x0, u_k = weight compensation; QMInitializer.cpp:33-41).
static int mpcSolveImpl(const qmgpu_problem* P, int N, double t0, const double* x0, const double* timeGrid, int K, const double* ttimes, const double* tstates,
                        const double* contactRef, int nEv, const double* ev, const int32_t* modes, const double* warmX, const double* warmU, int lineSearch,
                        double* outT, double* outX, double* outU, int32_t* outMode, double* stats) {
  Problem pr{P, inputWeight(*P), ModeSchedule{nEv, ev, modes}, Target{K, ttimes, tstates, contactRef}};
  std::vector<double> tg(N + 1);
  for (int k = 0; k <= N; ++k) tg[k] = timeGrid ? timeGrid[k] : t0 + k * P->settings.dt;
  std::vector<double> X((N + 1) * 30), U(N * 30);
  for (int k = 0; k <= N; ++k) for (int i = 0; i < 30; ++i) X[k * 30 + i] = warmX ? warmX[k * 30 + i] : x0[i];
  for (int i = 0; i < 30; ++i) X[i] = x0[i];
  for (int k = 0; k < N; ++k) {
    if (warmU) for (int i = 0; i < 30; ++i) U[k * 30 + i] = warmU[k * 30 + i];
    else weightCompensatingInput(*P, pr.ms.nodeModeAt(tg[k]), &U[k * 30]);
  }
  // sqp.sqpIteration iterations at most (task.info:77, 1 in the reference's configuration), each warm-started from the previous
  // iterate, with upstream's convergence test (SqpSolver::checkConvergence) after every one.
  SqpResult r = sqpIteration(pr, N, tg.data(), x0, X, U, lineSearch != 0);
  int iterations = 1, convergence = sqpConvergence(P->settings, 0, r);
  for (int it = 1; convergence == 0 && r.status == 0; ++it) {
    X = r.X; U = r.U;
    r = sqpIteration(pr, N, tg.data(), x0, X, U, lineSearch != 0);
    ++iterations; convergence = sqpConvergence(P->settings, it, r);
  }
  for (int k = 0; k <= N; ++k) { outT[k] = tg[k]; outMode[k] = pr.ms.nodeModeAt(tg[k]); }
  std::copy(r.X.begin(), r.X.end(), outX);
  std::copy(r.U.begin(), r.U.end(), outU);
  if (stats) { stats[0] = r.merit0; stats[1] = r.viol0; stats[2] = r.merit1; stats[3] = r.viol1; stats[4] = r.alpha; stats[5] = r.stepType; stats[6] = r.armijo; stats[7] = r.status; stats[8] = iterations; stats[9] = convergence; }
  return r.status;
}
int qmo_mpc_solve(const qmgpu_problem* P, int N, double t0, const double* x0, const double* timeGrid, int K, const double* ttimes, const double* tstates,
                  int nEv, const double* ev, const int32_t* modes, const double* warmX, const double* warmU, int lineSearch, double* outT, double* outX,
                  double* outU, int32_t* outMode, double* stats) {
  return mpcSolveImpl(P, N, t0, x0, timeGrid, K, ttimes, tstates, g_contactRef, nEv, ev, modes, warmX, warmU, lineSearch, outT, outX, outU, outMode, stats);
}

// MPC_MRT_Interface::evaluatePolicy as QMController::update uses it (qm_controllers/src/QMController.cpp:134-142): linear interpolation of the
// state / input trajectories at t (the input trajectory has N entries: its last one is held), planned mode = mode of the node interval holding t.
static void policyEval(int N, const double* tg, const double* X, const double* U, const int32_t* modes, double t, double* x, double* u, int32_t* mode) {
  int idx; double alpha;
  timeSegment(tg, N + 1, t, &idx, &alpha);
  for (int i = 0; i < 30; ++i) x[i] = alpha * X[idx * 30 + i] + (1.0 - alpha) * X[(idx + 1) * 30 + i];
  const int i0 = std::min(idx, N - 1), i1 = std::min(idx + 1, N - 1);
  for (int i = 0; i < 30; ++i) u[i] = alpha * U[i0 * 30 + i] + (1.0 - alpha) * U[i1 * 30 + i];
  int k = 0;
  while (k < N && tg[k + 1] < t) ++k;
  *mode = modes[k];
}

// The whole control cycle of a BATCH of independent instances, results returned -- what qmgpu_cycle_batch computes (MPC solve, policy evaluation at
// tEval, WBC update), one instance at a time on `threads` host threads (instance i on thread i % threads; every solve is self-contained).  The GPU
// parity tests use it to compare EVERY instance of the BASELINE configurations instead of a sample.  rbd == null: MPC only.
// Per-instance inputs: t0 [B] (null: 0), x0 [B][30], target knots [B][K] / [B][K][37], contact reference [B][K][6] or null, mode schedule
// nEv [B] / ev [B][QMGPU_MAX_EVENTS] / modes [B][QMGPU_MAX_EVENTS + 1]; tEval / period / time [B], rbd [B][55], inputLast [B][30] (in / out),
// eeForce [B][3] or null.  Returns the number of instances whose Riccati factorisation or WBC reported a failure.
int qmo_cycle_batch_warm_mt(const qmgpu_problem* P, int batch, int N, int K, int threads, const double* t0, const double* x0, const double* timeGrid /*[B][N+1] or null*/,
                            const double* warmX /*[B][N+1][30] or null*/, const double* warmU /*[B][N][30] or null*/, const double* ttimes, const double* tstates,
                       const double* contactRef, const int32_t* nEv, const double* ev, const int32_t* modes, int lineSearch, const double* tEval, const double* rbd,
                       const double* period, const double* time, double* inputLast, const double* eeForce, int variant, double* outX, double* outU,
                       int32_t* outMode, double* outStats, double* outPolicy /*[B][60] or null*/, int32_t* outPolicyMode /*[B] or null*/, double* outWbc /*[B][54]*/,
                       int32_t* outWbcStatus /*[B]*/) {
  if (threads < 1) threads = 1;
  mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 16 << 20);
  std::vector<int> failed(threads, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([=, &failed]() {
      std::vector<double> T(N + 1);
      for (int i = t; i < batch; i += threads) {
        double* Xi = outX + size_t(i) * (N + 1) * 30; double* Ui = outU + size_t(i) * N * 30; int32_t* Mi = outMode + size_t(i) * (N + 1);
        double* Si = outStats + size_t(i) * QMGPU_NSTATS;
        const int rc = mpcSolveImpl(P, N, t0 ? t0[i] : 0.0, x0 + size_t(i) * 30, timeGrid ? timeGrid + size_t(i) * (N + 1) : nullptr, K, ttimes + size_t(i) * K, tstates + size_t(i) * K * 37,
                                    contactRef ? contactRef + size_t(i) * K * 6 : nullptr, nEv[i], ev + size_t(i) * QMGPU_MAX_EVENTS,
                                    modes + size_t(i) * (QMGPU_MAX_EVENTS + 1), warmX ? warmX + size_t(i) * (N + 1) * 30 : nullptr, warmU ? warmU + size_t(i) * N * 30 : nullptr,
                                    lineSearch, T.data(), Xi, Ui, Mi, Si);
        int bad = rc != 0;
        if (rbd) {
          double xd[30], ud[30]; int32_t md;
          policyEval(N, T.data(), Xi, Ui, Mi, tEval[i], xd, ud, &md);
          if (outPolicy) for (int j = 0; j < 30; ++j) { outPolicy[size_t(i) * 60 + j] = xd[j]; outPolicy[size_t(i) * 60 + 30 + j] = ud[j]; }
          if (outPolicyMode) outPolicyMode[i] = md;
          const int ws = wbcUpdate(*P, variant, xd, ud, rbd + size_t(i) * 55, md, period[i], time[i], inputLast + size_t(i) * 30, outWbc + size_t(i) * 54, nullptr,
                                   eeForce ? eeForce + size_t(i) * 3 : nullptr, nullptr, g_wbcWorkingSet ? g_wbcWorkingSet + size_t(i) * QMGPU_WBC_STATE_WORDS : nullptr);
          outWbcStatus[i] = ws;
          bad = bad || ws != 0;
        }
        failed[t] += bad;
      }
    });
  for (auto& th : pool) th.join();
  int n = 0;
  for (int v : failed) n += v;
  return n;
}

int qmo_cycle_batch_mt(const qmgpu_problem* P, int batch, int N, int K, int threads, const double* t0, const double* x0, const double* ttimes, const double* tstates,
                       const double* contactRef, const int32_t* nEv, const double* ev, const int32_t* modes, int lineSearch, const double* tEval, const double* rbd,
                       const double* period, const double* time, double* inputLast, const double* eeForce, int variant, double* outX, double* outU,
                       int32_t* outMode, double* outStats, double* outPolicy, int32_t* outPolicyMode, double* outWbc, int32_t* outWbcStatus) {
  return qmo_cycle_batch_warm_mt(P, batch, N, K, threads, t0, x0, nullptr, nullptr, nullptr, ttimes, tstates, contactRef, nEv, ev, modes, lineSearch, tEval, rbd, period, time,
                                 inputLast, eeForce, variant, outX, outU, outMode, outStats, outPolicy, outPolicyMode, outWbc, outWbcStatus);
}

// experiment knobs of the WBC restatement (qmo_wbc.h; defaults = the product's algorithm): key 0 = starting value of the interior point that runs in front of the active-set method
// (default 300), key 3 = no interior point at all (the active-set method cold from z = 0 on every level), key 8 = the working sets carried from the previous tick are ignored, key 9 = per-iteration trace on stderr.  Both change the PATH to the
// vertex only: the tests use them to check that the result does not.
void qmo_set_experiment(int key, double value) {
  if (key == 0) g_expLowerLevelStart = value; else if (key == 3) g_expNoInteriorPoint = value != 0.0; else if (key == 4) g_expNoMinNormStart = value != 0.0; else if (key == 7) g_expGuessOrder = value != 0.0; else if (key == 8) g_expNoWarmStart = value != 0.0; else if (key == 12) g_expLiteralRegMaxN = int(value); else if (key == 13) g_expCanonicalFirst = int(value); else if (key == 14) g_expOwnInteriorPoint = int(value); else if (key == 15) g_expIpmStartDelta = value; else if (key == 9) g_expTrace = int(value);
}

// WBC updates of a BATCH of independent instances on `threads` host threads (what qmgpu_wbc_solve_batch computes): xDes / uDes [B][30], rbd [B][55],
// mode / period / time [B], inputLast [B][30] in / out, eeForce [B][3] or null, out [B][54], status [B], diag [B][8] or null (per level 0..3: re-solve attempts, interior-point iterations).  Returns the number of non-zero status words.
int qmo_wbc_batch_mt(const qmgpu_problem* P, int batch, int threads, int variant, const double* xDes, const double* uDes, const double* rbd, const int32_t* mode,
                     const double* period, const double* time, double* inputLast, const double* eeForce, double* out, int32_t* status, int32_t* diag /*[B][8] or null*/) {
  if (threads < 1) threads = 1;
  mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 16 << 20);
  std::vector<int> failed(threads, 0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([=, &failed]() {
      for (int i = t; i < batch; i += threads) {
        status[i] = wbcUpdate(*P, variant, xDes + size_t(i) * 30, uDes + size_t(i) * 30, rbd + size_t(i) * 55, mode[i], period[i], time[i], inputLast + size_t(i) * 30,
                              out + size_t(i) * 54, nullptr, eeForce ? eeForce + size_t(i) * 3 : nullptr, diag ? diag + size_t(i) * 8 : nullptr,
                              g_wbcWorkingSet ? g_wbcWorkingSet + size_t(i) * QMGPU_WBC_STATE_WORDS : nullptr);
        failed[t] += status[i] != 0;
      }
    });
  for (auto& th : pool) th.join();
  int n = 0;
  for (int v : failed) n += v;
  return n;
}

// Initial guess of the next solve from the previous solution (what upstream's SqpSolver::runImpl takes from its PrimalSolution; the counterpart of
// qmgpu_warm_start_batch): (X, U) of the previous grid evaluated at every node of the new grid with the policy-evaluation rule (end values held; the input
// trajectory has one entry less and holds its last value), x[0] replaced by x0.
void qmo_warm_start_batch(int batch, int Np, const double* gridP, const double* Xp, const double* Up, int Nn, const double* gridN, const double* x0, double* warmX,
                          double* warmU) {
  std::vector<int32_t> noModes(Np + 1, 0);
  for (int i = 0; i < batch; ++i)
    for (int k = 0; k <= Nn; ++k) {
      double x[30], u[30]; int32_t md;
      policyEval(Np, gridP + size_t(i) * (Np + 1), Xp + size_t(i) * (Np + 1) * 30, Up + size_t(i) * Np * 30, noModes.data(), gridN[size_t(i) * (Nn + 1) + k], x, u, &md);
      for (int j = 0; j < 30; ++j) warmX[(size_t(i) * (Nn + 1) + k) * 30 + j] = (k == 0 && x0) ? x0[size_t(i) * 30 + j] : x[j];
      if (k < Nn) for (int j = 0; j < 30; ++j) warmU[(size_t(i) * Nn + k) * 30 + j] = u[j];
    }
}

// policy evaluation of a batch (MPC_MRT_Interface::evaluatePolicy, see policyEval): T [B][N+1], X [B][N+1][30], U [B][N][30], modes [B][N+1], t [B] -> xu [B][60], mode [B]
void qmo_policy_eval_batch(int batch, int N, const double* T, const double* X, const double* U, const int32_t* modes, const double* t, double* xu, int32_t* modeOut) {
  for (int i = 0; i < batch; ++i)
    policyEval(N, T + size_t(i) * (N + 1), X + size_t(i) * (N + 1) * 30, U + size_t(i) * N * 30, modes + size_t(i) * (N + 1), t[i], xu + size_t(i) * 60, xu + size_t(i) * 60 + 30,
               modeOut + i);
}

// DDP variant (ddpIteration): warmU [N][30] or null -> the initializer's inputs; warmX [N+1][30] or null -> open-loop rollout of the inputs
int qmo_ddp_solve(const qmgpu_problem* P, int N, double t0, const double* x0, const double* timeGrid, int K, const double* ttimes, const double* tstates,
                  int nEv, const double* ev, const int32_t* modes, const double* warmX, const double* warmU, double* outT, double* outX, double* outU, int32_t* outMode,
                  double* stats) {
  Problem pr{P, inputWeight(*P), ModeSchedule{nEv, ev, modes}, Target{K, ttimes, tstates, g_contactRef}};
  std::vector<double> tg(N + 1), U(N * 30);
  for (int k = 0; k <= N; ++k) tg[k] = timeGrid ? timeGrid[k] : t0 + k * P->settings.dt;
  for (int k = 0; k < N; ++k) {
    if (warmU) for (int i = 0; i < 30; ++i) U[k * 30 + i] = warmU[k * 30 + i];
    else weightCompensatingInput(*P, pr.ms.nodeModeAt(tg[k]), &U[k * 30]);
  }
  const DdpResult r = ddpIteration(pr, N, tg.data(), x0, U, warmX);
  for (int k = 0; k <= N; ++k) { outT[k] = tg[k]; outMode[k] = pr.ms.nodeModeAt(tg[k]); }
  std::copy(r.X.begin(), r.X.end(), outX); std::copy(r.U.begin(), r.U.end(), outU);
  if (stats) { stats[0] = r.merit0; stats[1] = std::sqrt(r.eq0); stats[2] = r.merit1; stats[3] = std::sqrt(r.eq1); stats[4] = r.alpha; stats[5] = r.trials; stats[6] = r.armijo; stats[7] = r.status; stats[8] = 1; stats[9] = 1; }
  return r.status;
}

// performance index of a trajectory (merit, constraint violation)
void qmo_performance(const qmgpu_problem* P, int N, const double* tgrid, const double* x0, const double* X, const double* U, int K, const double* ttimes,
                     const double* tstates, int nEv, const double* ev, const int32_t* modes, double* merit, double* viol) {
  Problem pr{P, inputWeight(*P), ModeSchedule{nEv, ev, modes}, Target{K, ttimes, tstates, g_contactRef}};
  double cost = 0, dyn = 0, eq = 0;
  for (int i = 0; i < 30; ++i) dyn += (x0[i] - X[i]) * (x0[i] - X[i]);
  for (int k = 0; k < N; ++k) { const NodeMetrics m = nodeMetrics(pr, tgrid[k], tgrid[k + 1] - tgrid[k], X + k * 30, U + k * 30, X + (k + 1) * 30, false); cost += m.cost; dyn += m.dynViolationSSE; eq += m.eqViolationSSE; }
  cost += nodeMetrics(pr, tgrid[N], 0.0, X + N * 30, nullptr, nullptr, true).cost;
  *merit = cost; *viol = std::sqrt(dyn + eq);
}

// WBC model quantities for parity of the model-update kernel
void qmo_wbc_model(const qmgpu_problem* P, const double* xDes, const double* uDes, const double* rbd, double period, double* inputLast, double* M, double* nle,
                   double* J, double* dJ, double* baseJ, double* baseDJ, double* armJ, double* armDJ, double* qvMD /*4x24: qM vM qD vD*/, double* baseAcc,
                   double* feet /*4 x [posM velM posD velD] x3 = 48*/, double* ee /* posM velM posD velD angVelM (15) + RM(9) + RD(9) */) {
  WbcModel w;
  wbcUpdateMeasured(*P, rbd, w);
  wbcUpdateDesired(*P, xDes, uDes, inputLast, period, w);
  w.M.to(M); w.J.to(J); w.dJ.to(dJ); w.baseJ.to(baseJ); w.baseDJ.to(baseDJ); w.armJ.to(armJ); w.armDJ.to(armDJ);
  for (int i = 0; i < NV; ++i) { nle[i] = w.nle[i]; qvMD[i] = w.qM[i]; qvMD[NV + i] = w.vM[i]; qvMD[2 * NV + i] = w.qD[i]; qvMD[3 * NV + i] = w.vD[i]; }
  for (int i = 0; i < 6; ++i) baseAcc[i] = w.baseAccDesired[i];
  for (int c = 0; c < 4; ++c) for (int a = 0; a < 3; ++a) { feet[12 * c + a] = w.footPosM[c][a]; feet[12 * c + 3 + a] = w.footVelM[c][a]; feet[12 * c + 6 + a] = w.footPosD[c][a]; feet[12 * c + 9 + a] = w.footVelD[c][a]; }
  for (int a = 0; a < 3; ++a) { ee[a] = w.eePosM[a]; ee[3 + a] = w.eeVelM[a]; ee[6 + a] = w.eePosD[a]; ee[9 + a] = w.eeVelD[a]; ee[12 + a] = w.eeAngVelM[a]; }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { ee[15 + 3 * i + j] = w.eeRotM.m[i][j]; ee[24 + 3 * i + j] = w.eeRotD.m[i][j]; }
}

int qmo_wbc_update(const qmgpu_problem* P, int variant, const double* xDes, const double* uDes, const double* rbd, int mode, double period, double time,
                   double* inputLast, double* out54) {
  return wbcUpdate(*P, variant, xDes, uDes, rbd, mode, period, time, inputLast, out54, nullptr, g_wbcEeForce, nullptr, g_wbcWorkingSet);
}

// generic QP (row-major H n x n, D m x n) for KKT tests
int qmo_qp_solve(int n, int m, const double* H, const double* c, const double* D, const double* f, double* z, double* kktRes) {
  Mat Hm = Mat::from(H, n, n), Dm = m > 0 ? Mat::from(D, m, n) : Mat(0, n);
  Vec cv(c, c + n), fv(f, f + m), zv;
  const QpStats st = solveQpGeneric(Hm, cv, Dm, fv, zv);
  const int it = st.status ? -st.status : st.ipmIterations + st.iterations;
  if (kktRes) *kktRes = 0.0;
  for (int i = 0; i < n; ++i) z[i] = zv[i];
  return it;
}

// kernelFullPivLU on its own (tests: the pivot order is Eigen's).  A row major [rows][cols]; ker row major [cols][cols] (the first dimker columns are the basis);
// freeCols[dimker]: the original column each basis vector carries its 1 on; pivSeq[2 k], pivSeq[2 k + 1]: row / column POSITION of pivot k when it was chosen.  Returns dimker.
int qmo_kernel_full_piv_lu(int rows, int cols, const double* A, double* ker, int32_t* freeCols, int32_t* pivSeq, int32_t* nPiv) {
  std::vector<int> fr, seq;
  int rank = 0;
  const Mat N = kernelFullPivLU(Mat::from(A, rows, cols), &rank, &fr, &seq);
  for (int i = 0; i < cols; ++i) for (int j = 0; j < cols; ++j) ker[i * cols + j] = j < N.c ? N(i, j) : 0.0;
  for (size_t j = 0; j < fr.size(); ++j) freeCols[j] = fr[j];
  for (size_t k = 0; k < seq.size(); ++k) pivSeq[k] = seq[k];
  *nPiv = int(seq.size() / 2);
  return N.c;
}

// The three QPs of one WBC update (H c D f per level), for checking the GPU's structured solve against the dense data.
int qmo_wbc_levels(const qmgpu_problem* P, int variant, const double* xDes, const double* uDes, const double* rbd, int mode, double period, double time,
                   const double* inputLastIn, int level, int32_t* dims /*nz, rows, numDec*/, double* H, double* c, double* D, double* f, double* sol, double* xLevel) {
  double il[30]; for (int i = 0; i < 30; ++i) il[i] = inputLastIn[i];
  WbcModel w;
  wbcUpdateMeasured(*P, rbd, w);
  wbcUpdateDesired(*P, xDes, uDes, il, period, w);
  WbcTasks tk(*P, w, mode);
  const Task task0 = tk.floatingBaseEom() + tk.torqueLimits() + tk.noContactMotion() + tk.frictionCone();
  Task task1, task2;
  if (variant == 0) { task1 = (time < 10.0) ? tk.armJointNominalTracking() : (tk.baseHeight() + tk.baseAngular() + tk.eeLinear() + tk.eeAngular() + tk.swingLeg() * 100.0); task2 = tk.contactForce(uDes) + tk.baseLinear(); }
  else { task1 = tk.baseHeight() + tk.baseAngular() + tk.baseLinear() + tk.swingLeg() * 100.0; task2 = tk.contactForce(uDes); }
  HoQp h0(task0, nullptr);
  const HoQp* h = &h0;
  HoQp h1(task1, &h0);
  if (level >= 1) h = &h1;
  std::unique_ptr<HoQp> h2;
  if (level >= 2) { if (h1.Z.c == 0) return -1; h2.reset(new HoQp(task2, &h1)); h = h2.get(); }
  dims[0] = h->Hm.r; dims[1] = h->Dm.r; dims[2] = h->numDec;
  h->Hm.to(H); h->Dm.to(D);
  for (size_t i = 0; i < h->cv.size(); ++i) c[i] = h->cv[i];
  for (size_t i = 0; i < h->fv.size(); ++i) f[i] = h->fv[i];
  for (int i = 0; i < h->numDec; ++i) sol[i] = h->decSol[i];
  for (size_t i = 0; i < h->slackSol.size(); ++i) sol[h->numDec + i] = h->slackSol[i];
  const Vec x = h->solution();
  for (int i = 0; i < 36; ++i) xLevel[i] = x[i];
  return h->qpIters;
}

// Stacked task data (A b | D f) of one priority level, for debugging the task-assembly of the HIP kernel.
int qmo_wbc_task(const qmgpu_problem* P, int variant, const double* xDes, const double* uDes, const double* rbd, int mode, double period, double time,
                 const double* inputLastIn, int level, int32_t* dims /*ra, rd*/, double* A, double* b, double* D, double* f) {
  double il[30]; for (int i = 0; i < 30; ++i) il[i] = inputLastIn[i];
  WbcModel w;
  wbcUpdateMeasured(*P, rbd, w);
  wbcUpdateDesired(*P, xDes, uDes, il, period, w);
  WbcTasks tk(*P, w, mode);
  Task t;
  if (level == 0) t = tk.floatingBaseEom() + tk.torqueLimits() + tk.noContactMotion() + tk.frictionCone();
  else if (level == 1) {
    if (variant == 0) t = (time < 10.0) ? tk.armJointNominalTracking() : (tk.baseHeight() + tk.baseAngular() + tk.eeLinear() + tk.eeAngular() + tk.swingLeg() * 100.0);
    else t = tk.baseHeight() + tk.baseAngular() + tk.baseLinear() + tk.swingLeg() * 100.0;
  } else t = variant == 0 ? tk.contactForce(uDes) + tk.baseLinear() : tk.contactForce(uDes);
  dims[0] = t.a.r; dims[1] = t.d.r;
  if (t.a.r > 0) { t.a.to(A); for (int i = 0; i < t.a.r; ++i) b[i] = t.b[i]; }
  if (t.d.r > 0) { t.d.to(D); for (int i = 0; i < t.d.r; ++i) f[i] = t.f[i]; }
  return 0;
}

// CPU baseline: time `count` full MPC(+WBC) cycles, returns seconds.

// ---- front end of a control cycle
// x0: upstream CentroidalModelRbdConversions::computeCentroidalStateFromRbdModel as used at qm_controllers/src/QMController.cpp:239-244:
//     Pinocchio velocity v = [v_lin, ZYX Euler rates (from the world angular velocity), joint rates], x = [A(q) v / m ; q], yaw unwrapped.
// targets: qm_controllers/src/QmTargetTrajectoriesPublisher_node.cpp:59-254 and QMController.cpp:107-113 (kind 0).
void qmo_frontend(const qmgpu_problem* P, const double* rbd, double time, int haveYawLast, double yawLast, int kind, const double* cmd, double* lastEe, double feetHeight,
                  double armDist, double startX, double startY, double startPsi, double* x0, double* tt /*2*/, double* ts /*2x37*/) {
  const qmgpu_settings& st = P->settings;
  double q[NV], v[NV];
  for (int i = 0; i < 3; ++i) { q[i] = rbd[3 + i]; q[3 + i] = rbd[i]; v[i] = rbd[27 + i]; }
  for (int j = 0; j < 18; ++j) { q[6 + j] = rbd[6 + j]; v[6 + j] = rbd[30 + j]; }
  { const double sz = std::sin(q[3]), cz = std::cos(q[3]), sy = std::sin(q[4]), cy = std::cos(q[4]);
    const double wx = rbd[24], wy = rbd[25], wz = rbd[26], tmp = cz * wx / cy + sz * wy / cy;
    v[3] = sy * tmp + wz; v[4] = -sz * wx + cz * wy; v[5] = tmp; }
  Kin<double> k; forwardKinematics<double>(P->model, q, k);
  double Am[6][NV]; centroidalMomentumMatrix(P->model, k, Am);
  for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < NV; ++j) s += Am[i][j] * v[j]; x0[i] = s / P->model.total_mass; }
  for (int j = 0; j < NV; ++j) x0[6 + j] = q[j];
  if (haveYawLast) { double d = std::fmod(x0[9] - yawLast + M_PI, 2 * M_PI); if (d < 0) d += 2 * M_PI; x0[9] = yawLast + d - M_PI; }

  const double T = st.time_horizon, zRef = st.com_height + feetHeight;
  const double* ee = rbd + 48; const double* bc = x0 + 6;
  double s0[37] = {0}, s1[37] = {0}, tReach = time + T;
  for (int j = 0; j < 18; ++j) s0[12 + j] = s1[12 + j] = st.default_joint_state[j];
  s0[6] = bc[0]; s0[7] = bc[1]; s0[8] = zRef; s0[9] = bc[3];
  auto rotQ = [](const double* qq, const double* vv, double* o) {   // Eigen::Quaterniond::toRotationMatrix() * v, qq = (x, y, z, w)
    const double x = qq[0], y = qq[1], z = qq[2], w = qq[3];
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)}, {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    for (int i = 0; i < 3; ++i) o[i] = R[i][0] * vv[0] + R[i][1] * vv[1] + R[i][2] * vv[2];
  };
  if (kind == 1) {
    const M3<double> Rb = axisRotation<double>(2, bc[3]) * axisRotation<double>(1, bc[4]) * axisRotation<double>(0, bc[5]);   // getRotationMatrixFromZyxEulerAngles
    double vr[3]; for (int i = 0; i < 3; ++i) vr[i] = Rb.m[i][0] * cmd[0] + Rb.m[i][1] * cmd[1] + Rb.m[i][2] * cmd[2];
    s1[6] = bc[0] + vr[0] * T; s1[7] = bc[1] + vr[1] * T; s1[8] = zRef; s1[9] = bc[3] + cmd[3] * T;
    const double d = std::sqrt((lastEe[0] - ee[0]) * (lastEe[0] - ee[0]) + (lastEe[1] - ee[1]) * (lastEe[1] - ee[1]) + (lastEe[2] - ee[2]) * (lastEe[2] - ee[2]));
    if (d > 0.1) for (int i = 0; i < 3; ++i) lastEe[i] = ee[i];
    for (int i = 0; i < 7; ++i) s0[30 + i] = s1[30 + i] = lastEe[i];
    for (int i = 0; i < 3; ++i) s0[i] = s1[i] = vr[i];
  } else if (kind == 2) {
    const double qi[4] = {0, 0, -std::sin(bc[3] / 2), std::cos(bc[3] / 2)};
    double tmp[3], vr[3]; rotQ(qi, cmd, tmp); rotQ(ee + 3, tmp, vr);
    double e2[7]; for (int i = 0; i < 7; ++i) e2[i] = ee[i];
    e2[0] = ee[0] + vr[0] * T; e2[1] = ee[1] + vr[1] * T; e2[2] = lastEe[2]; e2[3] = lastEe[3]; e2[4] = lastEe[4];
    e2[5] = ee[5] + std::sin(vr[2] * T / 2); e2[6] = ee[6] + std::cos(vr[2] * T / 2);
    const double yaw = std::atan2(2.0 * (e2[6] * e2[5] + e2[3] * e2[4]), 1.0 - 2.0 * (e2[4] * e2[4] + e2[5] * e2[5]));
    s1[6] = e2[0] - armDist * std::cos(bc[3]); s1[7] = e2[1] - armDist * std::sin(bc[3]); s1[8] = zRef; s1[9] = yaw;
    for (int i = 0; i < 7; ++i) { s0[30 + i] = ee[i]; s1[30 + i] = e2[i]; }
  } else if (kind == 3) {
    const double yaw = std::atan2(2.0 * (cmd[6] * cmd[5] + cmd[3] * cmd[4]), 1.0 - 2.0 * (cmd[4] * cmd[4] + cmd[5] * cmd[5]));
    s1[6] = cmd[0] - armDist * std::cos(yaw); s1[7] = cmd[1] - armDist * std::sin(yaw); s1[8] = zRef; s1[9] = yaw;
    const V3<double> od = quaternionDistance<double>(ee + 3, cmd + 3);
    const double disp = std::sqrt((cmd[0] - ee[0]) * (cmd[0] - ee[0]) + (cmd[1] - ee[1]) * (cmd[1] - ee[1]) + (cmd[2] - ee[2]) * (cmd[2] - ee[2]));
    const double rot = std::sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]);
    tReach = time + std::max(rot / st.target_rotation_velocity, disp / st.target_displacement_velocity);
    for (int i = 0; i < 7; ++i) { s0[30 + i] = ee[i]; s1[30 + i] = cmd[i]; lastEe[i] = cmd[i]; }
  } else {
    for (int i = 0; i < 24; ++i) s0[i] = x0[i];
    for (int i = 0; i < 6; ++i) s0[24 + i] = st.initial_state[24 + i];
    s0[30] = startX + armDist * std::cos(startPsi); s0[31] = startY + armDist * std::sin(startPsi); s0[32] = st.com_height + rbd[5];
    s0[33] = 0; s0[34] = 0; s0[35] = std::sin(startPsi / 2); s0[36] = std::cos(startPsi / 2);
    for (int i = 0; i < 37; ++i) s1[i] = s0[i];
  }
  tt[0] = time; tt[1] = tReach;
  for (int i = 0; i < 37; ++i) { ts[i] = s0[i]; ts[37 + i] = s1[i]; }
}

double qmo_time_cycles(const qmgpu_problem* P, int count, int N, const double* x0s /*count x 30*/, int K, const double* ttimes, const double* tstates, int nEv,
                       const double* ev, const int32_t* modes, const double* rbds /*count x 55*/, int lineSearch) {
  std::vector<double> T(N + 1), X((N + 1) * 30), U(N * 30), st(QMGPU_NSTATS);
  std::vector<int32_t> md(N + 1);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < count; ++i) {
    qmo_mpc_solve(P, N, 0.0, x0s + i * 30, nullptr, K, ttimes, tstates, nEv, ev, modes, nullptr, nullptr, lineSearch, T.data(), X.data(), U.data(), md.data(), st.data());
    double il[30] = {0}, out[54];
    wbcUpdate(*P, 0, X.data(), U.data(), rbds + i * 55, md[0], 0.002, 20.0, il, out);
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// One instance at a time with `nodeThreads` worker threads over the shooting nodes: the reference's own configuration (task.info:78 nThreads = 3),
// SURVEY.md 8(d)'s second mode.  The WBC stays on the calling thread, as QMController::update runs it.
double qmo_time_cycles_node_threads(const qmgpu_problem* P, int count, int N, const double* x0s, int K, const double* ttimes, const double* tstates, int nEv,
                                    const double* ev, const int32_t* modes, const double* rbds, int lineSearch, int nodeThreads) {
  std::vector<double> X((N + 1) * 30), U(N * 30), tg(N + 1);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < count; ++i) {
    Problem pr{P, inputWeight(*P), ModeSchedule{nEv, ev, modes}, Target{K, ttimes, tstates, g_contactRef}};
    pr.nodeThreads = nodeThreads;
    for (int k = 0; k <= N; ++k) { tg[k] = k * P->settings.dt; for (int j = 0; j < 30; ++j) X[k * 30 + j] = x0s[i * 30 + j]; }
    for (int k = 0; k < N; ++k) weightCompensatingInput(*P, pr.ms.nodeModeAt(tg[k]), &U[k * 30]);
    const SqpResult r = sqpIteration(pr, N, tg.data(), x0s + i * 30, X, U, lineSearch != 0);
    double il[30] = {0}, out[54];
    wbcUpdate(*P, 0, r.X.data(), r.U.data(), rbds + i * 55, pr.ms.nodeModeAt(tg[0]), 0.002, 20.0, il, out);
  }
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Same, instances spread over `threads` host threads (thread t takes instances t, t + threads, ...): the "all hardware threads
// over instances" baseline of SURVEY.md 8(d).  Every solve is self-contained (thread_local scratch only).
double qmo_time_cycles_mt(const qmgpu_problem* P, int count, int N, const double* x0s, int K, const double* ttimes, const double* tstates, int nEv, const double* ev,
                          const int32_t* modes, const double* rbds, int lineSearch, int threads) {
  if (threads < 1) threads = 1;
  // Many threads of ONE process that return memory to the kernel (heap trims, munmap of large blocks) serialise on the process's address-space
  // lock: keep what was allocated (the oracle's matrices are short-lived std::vector storage) -- measured on the 256-thread GPU host.
  mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 16 << 20);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([=]() {
      std::vector<double> T(N + 1), X((N + 1) * 30), U(N * 30), st(QMGPU_NSTATS);
      std::vector<int32_t> md(N + 1);
      for (int i = t; i < count; i += threads) {
        qmo_mpc_solve(P, N, 0.0, x0s + i * 30, nullptr, K, ttimes, tstates, nEv, ev, modes, nullptr, nullptr, lineSearch, T.data(), X.data(), U.data(), md.data(), st.data());
        double il[30] = {0}, out[54];
        wbcUpdate(*P, 0, X.data(), U.data(), rbds + i * 55, md[0], 0.002, 20.0, il, out);
      }
    });
  for (auto& th : pool) th.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
