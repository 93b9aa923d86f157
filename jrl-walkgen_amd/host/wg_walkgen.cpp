// wg_walkgen.cpp -- host C++ facade (include/wg_walkgen.hh) over the C ABI.  No numerics here: every tick goes through
// wg_mpc_tick_batch (HIP); this file only carries the reference's plumbing (command dispatch, the four queues, the 5 ms
// clock) so that the reference's own test programs read the same against this library.
#include "../../include/wg_walkgen.hh"
#define WG_TRIG_FN static inline
#include "../../include/wg_trig.h"

#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

using namespace std;

namespace PatternGeneratorJRL {

// ---- pgtypes (src/pgtypes.cpp:27-70) ----------------------------------------------------------------------------------
COMPosition_s &COMPosition_s::operator=(const COMState_s &aCS) {
  for (unsigned int i = 0; i < 3; i++) { x[i] = aCS.x[i]; y[i] = aCS.y[i]; z[i] = aCS.z[i]; }
  yaw = aCS.yaw[0]; pitch = aCS.pitch[0]; roll = aCS.roll[0];
  return *this;
}
COMState_s &COMState_s::operator=(const COMPosition_s &aCS) {
  for (unsigned int i = 0; i < 3; i++) { x[i] = aCS.x[i]; y[i] = aCS.y[i]; z[i] = aCS.z[i]; }
  yaw[0] = aCS.yaw; yaw[1] = yaw[2] = 0.0;
  pitch[0] = aCS.pitch; pitch[1] = pitch[2] = 0.0;
  roll[0] = aCS.roll; roll[1] = roll[2] = 0.0;
  return *this;
}
void COMState_s::reset() {
  for (unsigned int i = 0; i < 3; i++) { x[i] = 0.0; y[i] = 0.0; yaw[i] = 0.0; pitch[i] = 0.0; roll[i] = 0.0; }
}
COMState_s::COMState_s() { reset(); }

HumanoidModel HumanoidModel::sampleRobot() {
  HumanoidModel h;
  memset(&h, 0, sizeof h);
  h.mass = 56.0;                               // only scales setCoMPerturbationForce, which the tick never reads
  h.soleWidth = 0.25; h.soleHeight = 0.14;
  h.hasHipYawLimits = false;
  h.startCoM[0] = 0.0316055; h.startCoM[1] = 0.0; h.startCoM[2] = 0.7116911;
  h.startLeftFoot[1] = 0.09; h.startRightFoot[1] = -0.09;
  return h;
}

static const double kPi = 3.14159265358979323846;

HumanoidModel HumanoidModelFromRobot(CjrlHumanoidDynamicRobot *aHDR) {
  if (aHDR == 0) throw runtime_error("HumanoidModelFromRobot: null robot");
  HumanoidModel h;
  memset(&h, 0, sizeof h);
  h.mass = aHDR->mass();                                   // ZMPVelocityReferencedQP.cpp:68
  CjrlFoot *RightFoot = aHDR->rightFoot(), *LeftFoot = aHDR->leftFoot();
  if (RightFoot == 0 || LeftFoot == 0) throw runtime_error("HumanoidModelFromRobot: the robot has no feet");
  double WidthHalf = 0.0, HeightHalf = 0.0;                // relative-feet-inequalities.cpp:155-172: right, then left; the
  RightFoot->getSoleSize(WidthHalf, HeightHalf);           // left foot's size ends up used for both feet
  LeftFoot->getSoleSize(WidthHalf, HeightHalf);
  h.soleWidth = WidthHalf; h.soleHeight = HeightHalf;
  vector3d ankle;
  LeftFoot->getAnklePositionInLocalFrame(ankle);           // rigid-body-system.cpp:38
  for (int i = 0; i < 3; i++) h.anklePosition[i] = ankle[i];
  // OrientationsPreview.cpp:42-68: hip-yaw joint = second joint on the chain waist -> ankle
  CjrlJoint *waist = aHDR->waist();
  const CjrlJoint *lAnkle = LeftFoot->associatedAnkle(), *rAnkle = RightFoot->associatedAnkle();
  if (waist && lAnkle && rAnkle) {
    vector<CjrlJoint *> lc = aHDR->jointsBetween(*waist, *lAnkle), rc = aHDR->jointsBetween(*waist, *rAnkle);
    if (lc.size() > 1 && rc.size() > 1) {
      h.hasHipYawLimits = true;
      h.leftHipYawLower = lc[1]->lowerBound(0); h.leftHipYawUpper = lc[1]->upperBound(0);
      h.rightHipYawLower = rc[1]->lowerBound(0); h.rightHipYawUpper = rc[1]->upperBound(0);
      if (h.leftHipYawLower == h.leftHipYawUpper) { h.leftHipYawLower = -30.0 / 180.0 * kPi; h.leftHipYawUpper = 45.0 / 180.0 * kPi; }
      if (h.rightHipYawLower == h.rightHipYawUpper) { h.rightHipYawLower = -30.0 / 180.0 * kPi; h.rightHipYawUpper = 45.0 / 180.0 * kPi; }
      h.hipYawVelocityMax = fabs(lc[1]->upperVelocityBound(0));
    }
  }
  return h;
}

// ---- SimplePluginManager / SimplePlugin (src/SimplePluginManager.cpp:40-163, src/SimplePlugin.cpp:35-50) -------------------
SimplePluginManager::~SimplePluginManager() {
  for (auto it = m_SimplePlugins.begin(); it != m_SimplePlugins.end(); ++it) it->second->m_SimplePluginManager = 0;
}
void SimplePluginManager::UnregisterPlugin(SimplePlugin *aSimplePlugin) {
  auto it = m_SimplePlugins.begin();
  while (it != m_SimplePlugins.end()) {
    if (it->second == aSimplePlugin) it = m_SimplePlugins.erase(it);
    else ++it;
  }
}
void SimplePluginManager::Print() {
  for (auto it = m_SimplePlugins.begin(); it != m_SimplePlugins.end(); ++it) cout << it->first << endl;
}
bool SimplePluginManager::RegisterMethod(string &MethodName, SimplePlugin *aSP) {
  m_SimplePlugins.insert(pair<string, SimplePlugin *>(MethodName, aSP));
  return true;
}
bool SimplePluginManager::CallMethod(string &MethodName, istringstream &istrm) {
  auto range = m_SimplePlugins.equal_range(MethodName);
  // every plugin registered for the method gets its own stream over the remaining arguments
  string rest;
  {
    streambuf *pbuf = istrm.rdbuf();
    streamsize size = pbuf->in_avail();
    for (streamsize i = 0; i < size; i++) rest.push_back((char)pbuf->sbumpc());
  }
  bool FoundAPlugin = false;
  for (auto it = range.first; it != range.second; ++it) {
    istringstream iss(rest);
    SimplePlugin *aSP = it->second;
    if (aSP != 0) { aSP->CallMethod(MethodName, iss); FoundAPlugin = true; }
    else cout << "SimplePlugin empty " << endl;
  }
  return FoundAPlugin;
}
bool SimplePlugin::RegisterMethod(string &MethodName) {
  bool r = false;
  if (m_SimplePluginManager != 0) r = m_SimplePluginManager->RegisterMethod(MethodName, this);
  return r;
}
SimplePlugin::~SimplePlugin() {
  if (m_SimplePluginManager != 0) m_SimplePluginManager->UnregisterPlugin(this);
}

// ---- ZMPRefTrajectoryGeneration (ZMPRefTrajectoryGeneration.cpp:34-110) -------------------------------------------------
ZMPRefTrajectoryGeneration::ZMPRefTrajectoryGeneration(SimplePluginManager *lSPM)
    : SimplePlugin(lSPM), m_Tsingle(0.), m_Tdble(0.), m_SamplingPeriod(0.), m_ModulationSupportCoefficient(0.), m_Omega(0.),
      m_PreviewControlTime(0.), m_StepHeight(0.), m_CurrentTime(0.), m_ComHeight(0.), m_OnLineMode(false) {
  string aMethodName[6] = {":omega", ":stepheight", ":singlesupporttime", ":doublesupporttime", ":comheight", ":samplingperiod"};
  for (int i = 0; i < 6; i++)
    if (!RegisterMethod(aMethodName[i])) cerr << "Unable to register " << aMethodName[i] << endl;
}
void ZMPRefTrajectoryGeneration::CallMethod(string &Method, istringstream &strm) {
  if (Method == ":omega") strm >> m_Omega;
  else if (Method == ":stepheight") strm >> m_StepHeight;
  else if (Method == ":singlesupporttime") strm >> m_Tsingle;
  else if (Method == ":doublesupporttime") strm >> m_Tdble;
  else if (Method == ":comheight") strm >> m_ComHeight;
  else if (Method == ":samplingperiod") strm >> m_SamplingPeriod;
}
bool ZMPRefTrajectoryGeneration::GetOnLineMode() { return m_OnLineMode; }

// ---- ZMPVelocityReferencedQP ------------------------------------------------------------------------------------------------
void solution_t::reset() { NbVariables = NbConstraints = Fail = NbIterations = NbActiveConstraints = 0; JerkX = JerkY = 0.0; }

static void wg_throw(const char *what) { throw runtime_error(string(what) + ": " + wg_last_error()); }
// The reference keeps its solver set-up per object; so does the facade: every object that configures device-side state
// (ZMPVelocityReferencedQP: model tables; PreviewControl: gains) owns a context of the C ABI.  WG_DEVICE picks the GPU.
static wg_ctx_t *NewContext() {
  const char *e = getenv("WG_DEVICE");
  wg_ctx_t *c = 0;
  if (wg_ctx_create(e ? atoi(e) : 0, &c) != WG_OK) wg_throw("wg_ctx_create");
  return c;
}

// ZMPVelocityReferencedQP.cpp:56-135
static void *NewPinnedBlock() {
  void *p = 0;
  if (wg_host_alloc(&p, sizeof(wg_gait_state_t) + sizeof(wg_tick_out_t) + 64) != WG_OK) wg_throw("wg_host_alloc");
  return p;
}
ZMPVelocityReferencedQP::ZMPVelocityReferencedQP(SimplePluginManager *SPM, string, const HumanoidModel *aHS)
    : ZMPRefTrajectoryGeneration(SPM), Pinned_(NewPinnedBlock()), State_(*static_cast<wg_gait_state_t *>(Pinned_)),
      Out_(reinterpret_cast<wg_tick_out_t *>(static_cast<wg_gait_state_t *>(Pinned_) + 1)) {
  if (aHS == 0) { wg_host_free(Pinned_); throw runtime_error("ZMPVelocityReferencedQP: a HumanoidModel is required"); }
  Running_ = false;
  Legacy_ = false;
  NbStepsSSDS_ = 2;                           // :79
  m_SamplingPeriod = 0.005;
  PerturbationOccured_ = false;
  for (int i = 0; i < 6; i++) PerturbationAcceleration_[i] = 0.0;
  RobotMass_ = aHS->mass;
  Solution_.reset();
  wg_model_defaults(&Model_);                 // QP_T_ 0.1, QP_N_ 16, 0.814, weights, FSM periods, OFTG constants
  Model_.sole_w = aHS->soleWidth;             // RelativeFeetInequalities::set_feet_dimensions
  Model_.sole_h = aHS->soleHeight;
  if (aHS->hasHipYawLimits) {                 // OrientationsPreview.cpp:42-68
    Model_.hip_l_lo = aHS->leftHipYawLower; Model_.hip_l_hi = aHS->leftHipYawUpper;
    Model_.hip_r_lo = aHS->rightHipYawLower; Model_.hip_r_hi = aHS->rightHipYawUpper;
    Model_.hip_vmax = fabs(aHS->hipYawVelocityMax);
  }
  memset(&State_, 0, sizeof State_);
  Ctx_ = NewContext();
  if (wg_mpc_configure_ctx(Ctx_, &Model_) != WG_OK) { wg_ctx_destroy(Ctx_); Ctx_ = 0; wg_host_free(Pinned_); wg_throw("wg_mpc_configure"); }
  // ":setfeetconstraint" is RelativeFeetInequalities' command in the reference (relative-feet-inequalities.cpp:68-79: of its
  // three names only this one is registered); that object's state is part of the device model here
  const unsigned int NbMethods = 4;
  string aMethodName[NbMethods] = {":previewcontroltime", ":numberstepsbeforestop", ":stoppg", ":setfeetconstraint"};
  for (unsigned int i = 0; i < NbMethods; i++)
    if (!RegisterMethod(aMethodName[i])) cerr << "Unable to register " << aMethodName[i] << endl;
}
ZMPVelocityReferencedQP::~ZMPVelocityReferencedQP() { wg_ctx_destroy(Ctx_); wg_host_free(Pinned_); }

void ZMPVelocityReferencedQP::setCoMPerturbationForce(istringstream &strm) {   // :162-173 (stored; no reader on the path)
  strm >> PerturbationAcceleration_[2];
  strm >> PerturbationAcceleration_[5];
  PerturbationAcceleration_[2] = PerturbationAcceleration_[2] / RobotMass_;
  PerturbationAcceleration_[5] = PerturbationAcceleration_[5] / RobotMass_;
  PerturbationOccured_ = true;
}
void ZMPVelocityReferencedQP::setCoMPerturbationForce(double x, double y) {     // :175-184
  PerturbationAcceleration_[2] = x / RobotMass_;
  PerturbationAcceleration_[5] = y / RobotMass_;
  PerturbationOccured_ = true;
}

void ZMPVelocityReferencedQP::CallMethod(string &Method, istringstream &strm) {   // :191-209
  if (Method == ":previewcontroltime") strm >> m_PreviewControlTime;
  if (Method == ":numberstepsbeforestop") {
    strm >> State_.nb_steps_left;                  // CurrentSupport.NbStepsLeft
    NbStepsSSDS_ = State_.nb_steps_left;           // SupportFSM_->NbStepsSSDS()
    State_.nb_steps_ssds = NbStepsSSDS_;
  }
  if (Method == ":stoppg") State_.ending_phase = 1;
  if (Method == ":setfeetconstraint") {             // relative-feet-inequalities.cpp:322-342
    string lCmd;
    strm >> lCmd;
    if (lCmd == "XY") {
      double mx = Model_.margin_x, my = Model_.margin_y;
      strm >> mx;
      strm >> my;
      SetFeetConstraint(mx, my);
      cout << "Security margin On X: " << mx << " Security margin On Y: " << my << endl;
    }
  }
  ZMPRefTrajectoryGeneration::CallMethod(Method, strm);
  if (Model_.Tctrl != m_SamplingPeriod) {           // ":samplingperiod": the device copy of the model follows (or refuses)
    const double old = Model_.Tctrl;
    Model_.Tctrl = m_SamplingPeriod;
    if (wg_mpc_configure_ctx(Ctx_, &Model_) != WG_OK) { Model_.Tctrl = old; m_SamplingPeriod = old; wg_throw("wg_mpc_configure"); }
  }
}

void ZMPVelocityReferencedQP::SetFeetConstraint(double SecurityMarginX, double SecurityMarginY) {
  const double ox = Model_.margin_x, oy = Model_.margin_y;
  Model_.margin_x = SecurityMarginX; Model_.margin_y = SecurityMarginY;
  if (wg_mpc_configure_ctx(Ctx_, &Model_) != WG_OK) { Model_.margin_x = ox; Model_.margin_y = oy; wg_throw("wg_mpc_configure"); }
}

void ZMPVelocityReferencedQP::LegacyGoldenReplay(bool on) {
  Legacy_ = on;
  if (on) Model_.flags |= WG_FLAG_NO_STOP_CENTERING;
  else Model_.flags &= ~WG_FLAG_NO_STOP_CENTERING;
  if (wg_mpc_configure_ctx(Ctx_, &Model_) != WG_OK) wg_throw("wg_mpc_configure");
}

static FootAbsolutePosition toFAP(const wg_foot_sample_t &s, double time, int stepType) {
  FootAbsolutePosition f;
  memcpy(&f, &s, sizeof s);           // same 18 leading doubles, pgtypes.hh:137-154
  f.time = time; f.stepType = stepType;
  return f;
}

// :212-319
int ZMPVelocityReferencedQP::InitOnLine(deque<ZMPPosition> &FinalZMPTraj_deq, deque<COMState> &FinalCoMPositions_deq,
                                        deque<FootAbsolutePosition> &FinalLeftFootTraj_deq,
                                        deque<FootAbsolutePosition> &FinalRightFootTraj_deq,
                                        FootAbsolutePosition &InitLeftFootAbsolutePosition,
                                        FootAbsolutePosition &InitRightFootAbsolutePosition, deque<RelativeFootPosition> &,
                                        COMState &lStartingCOMState, double lStartingZMPPosition[3]) {
  int AddArraySize;
  {
    assert(m_SamplingPeriod > 0);
    double ldAddArraySize = 0.04 / m_SamplingPeriod;        // TimeBuffer_ :62
    AddArraySize = (int)ldAddArraySize;
  }
  FinalZMPTraj_deq.resize(AddArraySize);
  FinalCoMPositions_deq.resize(AddArraySize);
  FinalLeftFootTraj_deq.resize(AddArraySize);
  FinalRightFootTraj_deq.resize(AddArraySize);
  m_CurrentTime = 0;
  for (unsigned int i = 0; i < FinalZMPTraj_deq.size(); i++) {
    FinalZMPTraj_deq[i].px = lStartingZMPPosition[0];
    FinalZMPTraj_deq[i].py = lStartingZMPPosition[1];
    FinalZMPTraj_deq[i].pz = lStartingZMPPosition[2];
    FinalZMPTraj_deq[i].theta = 0.0;
    FinalZMPTraj_deq[i].time = m_CurrentTime;
    FinalZMPTraj_deq[i].stepType = 0;
    FinalCoMPositions_deq[i] = lStartingCOMState;
    FinalLeftFootTraj_deq[i] = InitLeftFootAbsolutePosition;
    FinalRightFootTraj_deq[i] = InitRightFootAbsolutePosition;
    FinalLeftFootTraj_deq[i].time = FinalRightFootTraj_deq[i].time = m_CurrentTime;
    FinalLeftFootTraj_deq[i].stepType = FinalRightFootTraj_deq[i].stepType = 10;
    m_CurrentTime += m_SamplingPeriod;
  }
  const double com0[3] = {lStartingCOMState.x[0], lStartingCOMState.y[0], lStartingCOMState.z[0]};
  const double l[3] = {InitLeftFootAbsolutePosition.x, InitLeftFootAbsolutePosition.y, InitLeftFootAbsolutePosition.theta};
  const double r[3] = {InitRightFootAbsolutePosition.x, InitRightFootAbsolutePosition.y, InitRightFootAbsolutePosition.theta};
  const double vref[3] = {State_.vref[0], State_.vref[1], State_.vref[2]};
  wg_gait_init(&Model_, &State_, com0, l, r);           // support state, LIPM, trunk state, scheduling (:228-305)
  State_.com_x[1] = lStartingCOMState.x[1]; State_.com_x[2] = lStartingCOMState.x[2];
  State_.com_y[1] = lStartingCOMState.y[1]; State_.com_y[2] = lStartingCOMState.y[2];
  State_.vref[0] = vref[0]; State_.vref[1] = vref[1]; State_.vref[2] = vref[2];   // NewVelRef_ survives InitOnLine
  State_.nb_steps_ssds = NbStepsSSDS_;                                            // SupportFSM_ survives too
  if (Legacy_) State_.sup_y = 0.1;
  m_OnLineMode = true;
  Running_ = false;
  return 0;
}

// :323-458
void ZMPVelocityReferencedQP::OnLine(double time, deque<ZMPPosition> &FinalZMPTraj_deq, deque<COMState> &FinalCOMTraj_deq,
                                     deque<FootAbsolutePosition> &FinalLeftFootTraj_deq,
                                     deque<FootAbsolutePosition> &FinalRightFootTraj_deq) {
  if (!m_OnLineMode) return;
  if (State_.ending_phase && time >= State_.time_to_stop) { m_OnLineMode = false; State_.online = 0; }
  if (time + 0.00001 > State_.upper_time_limit) {
    State_.clock = time;
    const wg_tick_out_t &out = *Out_;
    const char *dump_failed = getenv("WG_DUMP_FAILED_QP"), *dump_every = getenv("WG_DUMP_EVERY_QP");
    wg_gait_state_t before;
    if (dump_failed || dump_every) before = State_;
    if (wg_mpc_tick_pinned_ctx(Ctx_, &State_, Out_, 0, 0) != WG_OK) wg_throw("wg_mpc_tick_pinned");
    Solution_.NbVariables = out.n; Solution_.NbConstraints = out.m; Solution_.Fail = out.ifail;
    Solution_.NbIterations = out.n_iter; Solution_.NbActiveConstraints = out.nact;
    Solution_.JerkX = out.jerk_x; Solution_.JerkY = out.jerk_y;
    if ((Solution_.Fail > 0 && dump_failed) || dump_every) {   // Problem_.dump( time ), ZMPVelocityReferencedQP.cpp:399-402
      const char *dir = dump_every ? dump_every : dump_failed;   // (the Herdt QP keeps jerk and foot placement free: it practically
      char Buffer[1024];                                         //  never fails -- WG_DUMP_EVERY_QP writes every tick's problem)
      snprintf(Buffer, sizeof Buffer, "%s/Problem_%f.dat", strcmp(dir, "1") == 0 ? "/tmp" : dir, time);
      dumpState(before, Buffer);
    }
    if (!FinalLeftFootTraj_deq.empty()) {       // the DS branch rewrites the newest queued sample, OFTG.cpp:333-336
      FinalLeftFootTraj_deq.back() = toFAP(out.lf_back, FinalLeftFootTraj_deq.back().time, FinalLeftFootTraj_deq.back().stepType);
      FinalRightFootTraj_deq.back() = toFAP(out.rf_back, FinalRightFootTraj_deq.back().time, FinalRightFootTraj_deq.back().stepType);
    }
    for (int k = 0; k < WG_SAMPLES_PER_TICK; ++k) {
      COMState c;
      for (int d = 0; d < 3; d++) { c.x[d] = out.com_x[k][d]; c.y[d] = out.com_y[k][d]; }
      c.z[0] = State_.com_z; c.z[1] = 0.0; c.z[2] = 0.0;
      c.yaw[0] = out.com_yaw[k][0]; c.yaw[1] = out.com_yaw[k][1];
      FinalCOMTraj_deq.push_back(c);
      ZMPPosition z;
      z.px = out.zmp_x[k]; z.py = out.zmp_y[k]; z.pz = 0.0; z.theta = 0.0;
      z.time = time + k * m_SamplingPeriod; z.stepType = 0;
      FinalZMPTraj_deq.push_back(z);
      FinalLeftFootTraj_deq.push_back(toFAP(out.lf[k], z.time, 0));
      FinalRightFootTraj_deq.push_back(toFAP(out.rf[k], z.time, 0));
    }
    Running_ = State_.running != 0;
  }
}

// QPProblem::dump_problem / dump(Type, aos) / dump_solver_parameters, qp-problem.cpp:548-675: names, shapes and layout as there
void ZMPVelocityReferencedQP::dumpState(const wg_gait_state_t &state, const char *FileName) {
  const int nmax = 2 * Model_.N + 2 * 4, mmax = 1 + 4 * Model_.N + 5 * 4 + 1;        // the largest problem a model of this horizon poses
  vector<double> Q((size_t)nmax * nmax), D(nmax), DU((size_t)mmax * nmax), DS(mmax), XL(nmax), XU(nmax);
  int n = 0, m = 0;
  if (wg_mpc_assemble_batch_ctx(Ctx_, 1, &state, 0, nmax, mmax, Q.data(), D.data(), DU.data(), DS.data(), XL.data(), XU.data(), &n, &m) != WG_OK)
    wg_throw("wg_mpc_assemble_batch");
  ofstream aof;
  aof.open(FileName, ofstream::out);
  if (!aof.is_open()) return;                              // the reference does not check either
  const int mmax_ = m + 1;                                 // QPProblem::solve: m_ = NbConstraints + 1, mmax_ = m_ + 1 (:249-251)
  auto mat = [&](const char *Name, const vector<double> &a, int ld, int NbRows, int NbCols) {
    aof << Name << "[" << NbRows << "," << NbCols << "]" << endl;
    for (int i = 0; i < NbRows; i++) {
      for (int j = 0; j < NbCols; j++) aof << a[(size_t)i + (size_t)j * ld] << " ";
      aof << endl;
    }
    aof << endl;
  };
  mat("Q", Q, nmax, n, n);
  mat("D", D, nmax, n, 1);
  mat("DU", DU, mmax, mmax_, n);
  mat("DS", DS, mmax, mmax_, 1);
  mat("XL", XL, nmax, n, 1);
  mat("XU", XU, nmax, n, 1);
  aof << "m: " << m << endl << "me: " << 0 << endl << "mmax: " << mmax_ << endl << "n: " << n << endl << "nmax: " << n << endl
      << "mnn: " << m + 2 * n << endl << "iout: " << 0 << endl << "iprint: " << 1 << endl
      << "lwar: " << 2 * (3 * n * n / 2 + 10 * n + 2 * m + 20000) << endl << "liwar: " << 2 * n + 1000 << endl << "Eps: " << 1e-8 << endl;
  aof.close();
}
void ZMPVelocityReferencedQP::dumpProblem(const char *FileName) {
  // the state's clock is the control-loop time of its last tick: the next tick fires one QP period later
  wg_gait_state_t next = State_;
  next.clock = State_.upper_time_limit;
  dumpState(next, FileName);
}
void ZMPVelocityReferencedQP::dumpProblem(double Time) {
  char Buffer[1024];
  snprintf(Buffer, sizeof Buffer, "/tmp/Problem_%f.dat", Time);
  dumpProblem(Buffer);
}

// ---- PatternGeneratorInterfacePrivate (Herdt branch) ------------------------------------------------------------------------
// ---- OptimalControllerSolver / PreviewControl ------------------------------------------------------------------------------
OptimalControllerSolver::OptimalControllerSolver(const vector<double> &A, const vector<double> &b, const vector<double> &c,
                                                 double Q, double R, unsigned int Nl)
    : m_A(A), m_b(b), m_c(c), m_Q(Q), m_R(R), m_Nl(Nl) {}

void OptimalControllerSolver::ComputeWeights(unsigned int Mode) {   // OptimalControllerSolver.cpp:200-352
  const int n = (int)m_b.size();
  m_K.assign(n, 0.0);
  m_F.assign(m_Nl, 0.0);
  const int mode = (Mode == MODE_WITHOUT_INITIALPOS) ? WG_RICCATI_WITHOUT_INITIALPOS : WG_RICCATI_WITH_INITIALPOS;
  if (wg_riccati_solve(n, m_A.data(), m_b.data(), m_c.data(), m_Q, m_R, (int)m_Nl, mode, m_K.data(), m_F.data()) != WG_OK)
    wg_throw("wg_riccati_solve");
}

PreviewControl::PreviewControl(SimplePluginManager *lSPM, unsigned int defaultMode, bool lAutoComputeWeights)
    : SimplePlugin(lSPM) {                                          // PreviewControl.cpp:37-77
  m_AutoComputeWeights = lAutoComputeWeights;
  m_DefaultWeightComputationMode = defaultMode;
  m_SamplingPeriod = 0.0; m_PreviewControlTime = 0.0; m_Zc = 0.0; m_SizeOfPreviewWindow = 0;
  m_Kx[0] = m_Kx[1] = m_Kx[2] = 0.0; m_Ks = 0;
  m_Coherent = false; m_Uploaded = false; m_Ctx = 0;
  string aMethodName[3] = {":samplingperiod", ":previewcontroltime", ":comheight"};
  for (int i = 0; i < 3; i++)
    if (!RegisterMethod(aMethodName[i])) cerr << "Unable to register " << aMethodName[i] << endl;
}
PreviewControl::~PreviewControl() { if (m_Ctx) wg_ctx_destroy(m_Ctx); }

void PreviewControl::SetSamplingPeriod(double v) {                  // :98-107
  if (m_SamplingPeriod != v) m_Coherent = false;
  m_SamplingPeriod = v;
  if (m_AutoComputeWeights) ComputeOptimalWeights(m_DefaultWeightComputationMode);
}
void PreviewControl::SetPreviewControlTime(double v) {              // :109-119
  if (m_PreviewControlTime != v) m_Coherent = false;
  m_PreviewControlTime = v;
  if (m_AutoComputeWeights) ComputeOptimalWeights(m_DefaultWeightComputationMode);
}
void PreviewControl::SetHeightOfCoM(double v) {                     // :121-131
  if (m_Zc != v) m_Coherent = false;
  m_Zc = v;
  if (m_AutoComputeWeights) ComputeOptimalWeights(m_DefaultWeightComputationMode);
}

void PreviewControl::ReadPrecomputedFile(string aFileName) {        // :138-194: the numbers pass through a float
  ifstream aif(aFileName.c_str(), ifstream::in);
  if (!aif.is_open()) { cerr << "PreviewControl - Unable to open " << aFileName << endl; return; }
  aif >> m_Zc; aif >> m_SamplingPeriod; aif >> m_PreviewControlTime;
  float r;
  for (int i = 0; i < 3; i++) { aif >> r; m_Kx[i] = r; }
  aif >> r; m_Ks = r;
  m_SizeOfPreviewWindow = (unsigned int)(m_PreviewControlTime / m_SamplingPeriod);
  m_F.assign(m_SizeOfPreviewWindow, 0.0);
  for (unsigned int i = 0; i < m_SizeOfPreviewWindow; i++) { aif >> r; m_F[i] = r; }
  m_Coherent = true; m_Uploaded = false;
}

void PreviewControl::ComputeOptimalWeights(unsigned int mode) {     // :198-322
  const double T = m_SamplingPeriod;
  if (T == 0.0) return;
  if (m_PreviewControlTime == 0.0) return;
  const int Nl = (int)(m_PreviewControlTime / T);
  m_F.assign(Nl, 0.0);
  double K[4] = {0, 0, 0, 0};
  if (mode == OptimalControllerSolver::MODE_WITHOUT_INITIALPOS) {
    if (wg_riccati_gains(T, m_Zc, 1.0, 1e-6, Nl, WG_RICCATI_WITHOUT_INITIALPOS, K, m_F.data()) != WG_OK) wg_throw("wg_riccati_gains");
    m_Ks = K[0]; m_Kx[0] = K[1]; m_Kx[1] = K[2]; m_Kx[2] = K[3];
  } else if (mode == OptimalControllerSolver::MODE_WITH_INITIALPOS) {
    if (wg_riccati_gains(T, m_Zc, 1.0, 1e-5, Nl, WG_RICCATI_WITH_INITIALPOS, K, m_F.data()) != WG_OK) wg_throw("wg_riccati_gains");
    m_Ks = K[0]; m_Kx[0] = K[0]; m_Kx[1] = K[1]; m_Kx[2] = K[2];
  }
  m_SizeOfPreviewWindow = (unsigned int)(m_PreviewControlTime / m_SamplingPeriod);
  m_F.resize(m_SizeOfPreviewWindow);
  m_Coherent = true; m_Uploaded = false;
}

void PreviewControl::Upload() {
  if (m_Uploaded) return;
  wg_preview_gains_t g;
  memset(&g, 0, sizeof g);
  g.T = m_SamplingPeriod; g.zc = m_Zc; g.Ks = m_Ks; g.Kx[0] = m_Kx[0]; g.Kx[1] = m_Kx[1]; g.Kx[2] = m_Kx[2];
  g.nl = (int)m_SizeOfPreviewWindow;
  if (!m_Ctx) m_Ctx = NewContext();
  if (wg_preview_configure_ctx(m_Ctx, &g, m_F.data()) != WG_OK) wg_throw("wg_preview_configure");
  m_Uploaded = true;
}

int PreviewControl::RunBatch(int B, int L, const double *zmp_x, const double *zmp_y, double *state, double *com, double *zmp2,
                             bool Simulation) {
  Upload();
  if (wg_preview_run_batch_ctx(m_Ctx, B, L, zmp_x, zmp_y, state, com, zmp2, Simulation ? 1 : 0) != WG_OK) wg_throw("wg_preview_run_batch");
  return 0;
}

int PreviewControl::OneIterationOfPreview(vector<double> &x, vector<double> &y, double &sxzmp, double &syzmp,
                                          deque<ZMPPosition> &ZMPPositions, unsigned int lindex, double &zmpx2, double &zmpy2,
                                          bool Simulation) {         // :324-374
  const unsigned int nl = m_SizeOfPreviewWindow;
  if (ZMPPositions.size() < nl) throw runtime_error("ZMPPositions.size()<m_SizeOfPreviewWindow:");   // LTHROW, :341-344
  if (ZMPPositions.size() < lindex + nl) throw runtime_error("PreviewControl: preview window runs past the ZMP queue");
  vector<double> zx(nl), zy(nl);
  for (unsigned int i = 0; i < nl; i++) { zx[i] = ZMPPositions[lindex + i].px; zy[i] = ZMPPositions[lindex + i].py; }
  double st[8] = {x[0], x[1], x[2], y[0], y[1], y[2], sxzmp, syzmp}, z2[2] = {0, 0};
  RunBatch(1, 1, zx.data(), zy.data(), st, 0, z2, Simulation);
  for (int i = 0; i < 3; i++) { x[i] = st[i]; y[i] = st[3 + i]; }
  sxzmp = st[6]; syzmp = st[7]; zmpx2 = z2[0]; zmpy2 = z2[1];
  return 0;
}

template <class Q>
static int preview_1d(PreviewControl &pc, unsigned int nl, vector<double> &x, double &sxzmp, Q &ZMPPositions, unsigned int lindex,
                      double &zmpx2, bool Simulation) {              // :376-420, :422-...
  if (ZMPPositions.size() < nl) throw runtime_error("ZMPPositions.size()<m_SizeOfPreviewWindow");   // the reference exit(0)s here
  vector<double> zx(nl), zy(nl, 0.0);
  if (ZMPPositions.size() >= lindex + nl)
    for (unsigned int i = 0; i < nl; i++) zx[i] = ZMPPositions[lindex + i];
  else
    throw runtime_error("PreviewControl: preview window runs past the ZMP queue");
  double st[8] = {x[0], x[1], x[2], 0, 0, 0, sxzmp, 0}, z2[2] = {0, 0};
  pc.RunBatch(1, 1, zx.data(), zy.data(), st, 0, z2, Simulation);
  for (int i = 0; i < 3; i++) x[i] = st[i];
  sxzmp = st[6]; zmpx2 = z2[0];
  return 0;
}
int PreviewControl::OneIterationOfPreview1D(vector<double> &x, double &sxzmp, deque<double> &ZMPPositions, unsigned int lindex,
                                            double &zmpx2, bool Simulation) {
  return preview_1d(*this, m_SizeOfPreviewWindow, x, sxzmp, ZMPPositions, lindex, zmpx2, Simulation);
}
int PreviewControl::OneIterationOfPreview1D(vector<double> &x, double &sxzmp, vector<double> &ZMPPositions, unsigned int lindex,
                                            double &zmpx2, bool Simulation) {
  return preview_1d(*this, m_SizeOfPreviewWindow, x, sxzmp, ZMPPositions, lindex, zmpx2, Simulation);
}

void PreviewControl::print() {                                       // :470-...
  cout << "Zc: " << m_Zc << " T: " << m_SamplingPeriod << " preview time: " << m_PreviewControlTime << endl;
  cout << "Ks: " << m_Ks << " Kx: " << m_Kx[0] << " " << m_Kx[1] << " " << m_Kx[2] << endl;
  for (size_t i = 0; i < m_F.size(); i++) cout << "F[" << i << "]: " << m_F[i] << endl;
}

void PreviewControl::CallMethod(string &Method, istringstream &astrm) {   // :515-540
  if (Method == ":samplingperiod") { string aws; if (astrm.good()) { double x; astrm >> x; SetSamplingPeriod(x); } }
  else if (Method == ":previewcontroltime") { if (astrm.good()) { double x; astrm >> x; SetPreviewControlTime(x); } }
  else if (Method == ":comheight") { if (astrm.good()) { double x; astrm >> x; SetHeightOfCoM(x); } }
  else if (Method == ":computeweightsofpreview") {
    if (astrm.good()) {
      string initialpos;
      astrm >> initialpos;
      if (initialpos == "withinitialpos") ComputeOptimalWeights(OptimalControllerSolver::MODE_WITH_INITIALPOS);
      else if (initialpos == "withoutinitialpos") ComputeOptimalWeights(OptimalControllerSolver::MODE_WITHOUT_INITIALPOS);
    }
  }
}

// ---- ZMPDiscretization ------------------------------------------------------------------------------------------------------
// ZMPDiscretization.cpp:78-131
ZMPDiscretization::ZMPDiscretization(SimplePluginManager *lSPM, string, const HumanoidModel *aHS)
    : ZMPRefTrajectoryGeneration(lSPM) {
  wg_zmpdisc_defaults(&Model_);
  m_ModulationSupportCoefficient = 0.9;      // :99
  m_SamplingPeriod = Model_.T; m_PreviewControlTime = Model_.preview_time;
  m_Tsingle = Model_.t_single; m_Tdble = Model_.t_double; m_StepHeight = Model_.step_height; m_Omega = Model_.omega;
  if (aHS) {                                 // FootTrajectoryGenerationStandard.cpp:62-71: lDepth is the ankle's height
    Model_.foot_b = aHS->anklePosition[0];
    Model_.foot_h = aHS->anklePosition[2];
    Model_.foot_f = aHS->anklePosition[2] - aHS->anklePosition[0];
  }
  for (int i = 0; i < 6; i++) InitFeet_[i] = 0.0;
  StartTime_ = 0.0;
  Produced_ = 0;
  string aMethodName[3] = {":prevzmpinitprofil", ":zeroinitprofil", ":previewcontroltime"};   // :1302-1322
  for (int i = 0; i < 3; i++)
    if (!RegisterMethod(aMethodName[i])) cerr << "Unable to register " << aMethodName[i] << endl;
}
ZMPDiscretization::~ZMPDiscretization() {}

void ZMPDiscretization::CallMethod(string &Method, istringstream &strm) {   // :1323-1345
  if (Method == ":previewcontroltime") strm >> m_PreviewControlTime;
  ZMPRefTrajectoryGeneration::CallMethod(Method, strm);
}
void ZMPDiscretization::SetZMPShift(vector<double> &ZMPShift) {             // :305-314
  for (size_t i = 0; i < ZMPShift.size() && i < 4; i++) Model_.zmp_shift[i] = ZMPShift[i];
}
int ZMPDiscretization::ReturnOptimalTimeToRegenerateAStep() {               // :1122-1127
  return 2 * (int)(m_PreviewControlTime / m_SamplingPeriod);
}
const wg_zmpdisc_model_t &ZMPDiscretization::Model() {
  Model_.T = m_SamplingPeriod; Model_.preview_time = m_PreviewControlTime;
  Model_.t_single = m_Tsingle; Model_.t_double = m_Tdble; Model_.step_height = m_StepHeight; Model_.omega = m_Omega;
  Model_.modulation = m_ModulationSupportCoefficient;
  return Model_;
}

void ZMPDiscretization::Produce(bool with_end, size_t from, deque<ZMPPosition> &Z, deque<COMState> &Cs,
                                deque<FootAbsolutePosition> &L, deque<FootAbsolutePosition> &R) {
  const wg_zmpdisc_model_t &M = Model();
  const int S = (int)Steps_.size();
  const int total = wg_zmpdisc_length(&M, Steps_.data(), S);
  if (total < 0) throw runtime_error("ZMPDiscretization: unsupported step sequence");
  const size_t n_end = (size_t)((unsigned)round(M.t_double / (2 * M.T))) + (size_t)(int)(3.0 * M.preview_time / M.T);
  const size_t stop = with_end ? (size_t)total : (size_t)total - n_end;
  vector<double> zmp((size_t)total * 2), zth(total), lf((size_t)total * 6), rf((size_t)total * 6);
  vector<int> zty(total), lty(total), rty(total);
  int len = 0;
  if (wg_zmpdisc_batch(&M, 1, S, Steps_.data(), &S, InitFeet_, total, zmp.data(), zth.data(), zty.data(), lf.data(), lty.data(),
                       rf.data(), rty.data(), &len) != WG_OK)
    wg_throw("wg_zmpdisc_batch");
  if (len != total) throw runtime_error("ZMPDiscretization: the step sequence was refused");
  double t = StartTime_;                     // m_CurrentTime += m_SamplingPeriod per sample
  for (size_t i = 0; i < stop; i++, t += M.T) {
    if (i < from) continue;
    ZMPPosition z;
    memset(&z, 0, sizeof z);
    z.px = zmp[2 * i]; z.py = zmp[2 * i + 1]; z.theta = zth[i]; z.time = t; z.stepType = zty[i];
    Z.push_back(z);
    COMState c;
    c.z[0] = m_ComHeight;
    c.yaw[0] = zth[i];
    Cs.push_back(c);
    for (int side = 0; side < 2; side++) {
      const double *f = (side ? rf.data() : lf.data()) + 6 * i;
      FootAbsolutePosition a;
      memset(&a, 0, sizeof a);
      a.x = f[0]; a.y = f[1]; a.z = f[2]; a.theta = f[3]; a.omega = f[4]; a.omega2 = f[5];
      a.time = t; a.stepType = side ? rty[i] : lty[i];
      (side ? R : L).push_back(a);
    }
  }
  if (!with_end) Produced_ = stop;
  m_CurrentTime = t;
}

// :319-513
int ZMPDiscretization::InitOnLine(deque<ZMPPosition> &FinalZMPPositions, deque<COMState> &COMStates,
                                  deque<FootAbsolutePosition> &FinalLeftFootAbsolutePositions,
                                  deque<FootAbsolutePosition> &FinalRightFootAbsolutePositions,
                                  FootAbsolutePosition &InitLeftFootAbsolutePosition,
                                  FootAbsolutePosition &InitRightFootAbsolutePosition,
                                  deque<RelativeFootPosition> &RelativeFootPositions, COMState &, double *) {
  if (RelativeFootPositions.size() < 2) throw runtime_error("ZMPDiscretization::InitOnLine: at least two steps are needed");
  Steps_.clear();
  for (size_t i = 0; i < RelativeFootPositions.size(); i++) {
    const RelativeFootPosition &r = RelativeFootPositions[i];
    wg_rel_step_t s;
    memset(&s, 0, sizeof s);
    s.sx = r.sx; s.sy = r.sy; s.theta = r.theta; s.ss_time = r.SStime; s.ds_time = r.DStime; s.step_type = r.stepType;
    Steps_.push_back(s);
  }
  InitFeet_[0] = InitLeftFootAbsolutePosition.x; InitFeet_[1] = InitLeftFootAbsolutePosition.y;
  InitFeet_[2] = InitLeftFootAbsolutePosition.theta;
  InitFeet_[3] = InitRightFootAbsolutePosition.x; InitFeet_[4] = InitRightFootAbsolutePosition.y;
  InitFeet_[5] = InitRightFootAbsolutePosition.theta;
  StartTime_ = m_CurrentTime;
  Produced_ = 0;
  Produce(false, 0, FinalZMPPositions, COMStates, FinalLeftFootAbsolutePositions, FinalRightFootAbsolutePositions);
  return (int)RelativeFootPositions.size();
}

void ZMPDiscretization::OnLine(double, deque<ZMPPosition> &, deque<COMState> &, deque<FootAbsolutePosition> &,
                               deque<FootAbsolutePosition> &) {}   // :561-568: does nothing

// :573-1020
void ZMPDiscretization::OnLineAddFoot(RelativeFootPosition &r, deque<ZMPPosition> &FinalZMPPositions, deque<COMState> &COMStates,
                                      deque<FootAbsolutePosition> &FinalLeftFootAbsolutePositions,
                                      deque<FootAbsolutePosition> &FinalRightFootAbsolutePositions, bool EndSequence) {
  if (Steps_.empty()) throw runtime_error("ZMPDiscretization::OnLineAddFoot before InitOnLine");
  wg_rel_step_t s;
  memset(&s, 0, sizeof s);
  s.sx = r.sx; s.sy = r.sy; s.theta = r.theta; s.ss_time = r.SStime; s.ds_time = r.DStime; s.step_type = r.stepType;
  Steps_.push_back(s);
  Produce(EndSequence, Produced_, FinalZMPPositions, COMStates, FinalLeftFootAbsolutePositions, FinalRightFootAbsolutePositions);
}

// :1129-1300
void ZMPDiscretization::EndPhaseOfTheWalking(deque<ZMPPosition> &ZMPPositions, deque<COMState> &FinalCOMStates,
                                             deque<FootAbsolutePosition> &LeftFootAbsolutePositions,
                                             deque<FootAbsolutePosition> &RightFootAbsolutePositions) {
  if (Steps_.empty()) throw runtime_error("ZMPDiscretization::EndPhaseOfTheWalking before InitOnLine");
  Produce(true, Produced_, ZMPPositions, FinalCOMStates, LeftFootAbsolutePositions, RightFootAbsolutePositions);
}

// :143-173
void ZMPDiscretization::GetZMPDiscretization(deque<ZMPPosition> &FinalZMPPositions, deque<COMState> &FinalCOMStates,
                                             deque<RelativeFootPosition> &RelativeFootPositions,
                                             deque<FootAbsolutePosition> &LeftFootAbsolutePositions,
                                             deque<FootAbsolutePosition> &RightFootAbsolutePositions, double,
                                             COMState &lStartingCOMState, double lStartingZMPPosition[3],
                                             FootAbsolutePosition &InitLeftFootAbsolutePosition,
                                             FootAbsolutePosition &InitRightFootAbsolutePosition) {
  InitOnLine(FinalZMPPositions, FinalCOMStates, LeftFootAbsolutePositions, RightFootAbsolutePositions,
             InitLeftFootAbsolutePosition, InitRightFootAbsolutePosition, RelativeFootPositions, lStartingCOMState,
             lStartingZMPPosition);
  EndPhaseOfTheWalking(FinalZMPPositions, FinalCOMStates, LeftFootAbsolutePositions, RightFootAbsolutePositions);
  FinalCOMStates.resize(FinalZMPPositions.size());
}

// ---- StepStackHandler (walk mode 0), StepStackHandler.cpp:128-175 -------------------------------------------------------------
void StepStackHandler::ReadStepSequenceAccordingToWalkMode(istringstream &strm) {
  m_RelativeFootPositions.clear();
  RelativeFootPosition aFootPosition;
  memset(&aFootPosition, 0, sizeof aFootPosition);
  // a token that is not a number sets failbit without ever reaching eof: the extraction's own result ends the loop
  // (well-formed input takes the same path as the reference's eof tests)
  while (!strm.eof()) {
    if (strm.eof() || !(strm >> aFootPosition.sx)) break;
    if (strm.eof() || !(strm >> aFootPosition.sy)) break;
    if (strm.eof() || !(strm >> aFootPosition.theta)) break;
    aFootPosition.DeviationHipHeight = 0;
    aFootPosition.SStime = m_SingleSupportTime;
    aFootPosition.DStime = m_DoubleSupportTime;
    aFootPosition.stepType = 1;
    m_RelativeFootPositions.push_back(aFootPosition);
    m_KeepLastCorrectSupportFoot = aFootPosition.sy > 0 ? -1 : 1;
  }
}
void StepStackHandler::AddStepInTheStack(double sx, double sy, double theta, double sstime, double dstime) {   // :850-863
  RelativeFootPosition aFootPosition;
  memset(&aFootPosition, 0, sizeof aFootPosition);
  aFootPosition.sx = sx; aFootPosition.sy = sy; aFootPosition.theta = theta;
  aFootPosition.SStime = sstime; aFootPosition.DStime = dstime; aFootPosition.stepType = 0;
  m_RelativeFootPositions.push_back(aFootPosition);
}
static void push_rel(deque<RelativeFootPosition> &q, double sx, double sy, double theta, double ss, double ds) {
  RelativeFootPosition a;
  memset(&a, 0, sizeof a);               // the reference leaves stepType uninitialised in the first two generators; 0 here
  a.sx = sx; a.sy = sy; a.theta = theta; a.SStime = ss; a.DStime = ds; a.stepType = 0;
  q.push_back(a);
}
void StepStackHandler::PrepareForSupportFoot(int SupportFoot) {            // :754-764
  push_rel(m_RelativeFootPositions, 0, SupportFoot * 0.095, 0, m_SingleSupportTime, m_DoubleSupportTime);
}
void StepStackHandler::FinishOnTheLastCorrectSupportFoot() {               // :872-882
  push_rel(m_RelativeFootPositions, 0, m_KeepLastCorrectSupportFoot * 0.19, 0, m_SingleSupportTime, m_DoubleSupportTime);
}
// :299-457: steps of 0.15 m along an arc of radius |(x, y)|, the last one shorter; each step is expressed in the frame of
// the previous footprint (rotation by the accumulated heading, transposed)
void StepStackHandler::CreateArcInStepStack(double x, double y, double R, double arc_deg, int SupportFoot) {
  const double kPi = 3.14159265358979323846;
  double StepMax = 0.15;
  const double OmegaTotal = arc_deg * kPi / 180.0;
  int DirectionRay = -1;
  R = sqrt(x * x + y * y);
  const int NumberOfStep = (int)floor(OmegaTotal * R / StepMax);
  double LastStep = OmegaTotal * R - NumberOfStep * StepMax;
  double OmegaStep = StepMax / R;
  double LastOmegaStep = OmegaTotal - OmegaStep * NumberOfStep;
  OmegaStep = OmegaStep * 180.0 / kPi;
  LastOmegaStep = LastOmegaStep * 180.0 / kPi;
  if (x < 0) { StepMax = -StepMax; LastStep = -LastStep; DirectionRay = 1; }
  if (y < 0) { OmegaStep = -OmegaStep; LastOmegaStep = -LastOmegaStep; }
  double Omegak = 0.0;
  for (int i = 0; i <= NumberOfStep; i++) {
    const bool last = i == NumberOfStep;
    if (last && LastStep == 0.0) break;
    const double dOmega = last ? LastOmegaStep : OmegaStep;
    const double Omegakp = Omegak;
    Omegak = Omegak + dOmega;
    const double c = wg_cos(Omegak * kPi / 180.0), s = wg_sin(Omegak * kPi / 180.0);
    const double cp = wg_cos(Omegakp * kPi / 180.0), sp = wg_sin(Omegakp * kPi / 180.0);
    const double outer = R + DirectionRay * SupportFoot * 0.095, inner = R - DirectionRay * SupportFoot * 0.095;
    const double lv0 = outer * s - inner * sp, lv1 = -(outer * c - inner * cp);
    double sx = 0.0, sy = 0.0;
    sx += c * lv0; sx += s * lv1;
    sy += -s * lv0; sy += c * lv1;
    push_rel(m_RelativeFootPositions, sx, sy, dOmega, m_SingleSupportTime, m_DoubleSupportTime);
    SupportFoot = -SupportFoot;
  }
  m_KeepLastCorrectSupportFoot = SupportFoot;
}
// :459-752: steps of 0.10 m of arc around a centre at distance R to the side, each pair (arc step, closing step) built from
// homogeneous 2-D transforms: where the non-support foot must land on the rotated radius, expressed in the frame of the current
// support footprint.  The reference inverts its 3 x 3 matrices numerically (MAL_INVERSE); they are rigid transforms, inverted
// here in closed form ([R' | -R't]) -- no golden file of the reference reaches this generator (parity unpinned).
namespace {
struct M3 {
  double a[3][3];
  static M3 rigid(double angle, double tx, double ty) {
    M3 m;
    const double c = wg_cos(angle), s = wg_sin(angle);
    m.a[0][0] = c; m.a[0][1] = -s; m.a[0][2] = tx;
    m.a[1][0] = s; m.a[1][1] = c;  m.a[1][2] = ty;
    m.a[2][0] = 0; m.a[2][1] = 0;  m.a[2][2] = 1;
    return m;
  }
  M3 operator*(const M3 &o) const {
    M3 r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double t = 0.0;
        for (int k = 0; k < 3; k++) t += a[i][k] * o.a[k][j];
        r.a[i][j] = t;
      }
    return r;
  }
  M3 rigid_inverse() const {
    M3 r;
    r.a[0][0] = a[0][0]; r.a[0][1] = a[1][0]; r.a[1][0] = a[0][1]; r.a[1][1] = a[1][1];
    r.a[0][2] = -(a[0][0] * a[0][2] + a[1][0] * a[1][2]);
    r.a[1][2] = -(a[0][1] * a[0][2] + a[1][1] * a[1][2]);
    r.a[2][0] = 0; r.a[2][1] = 0; r.a[2][2] = 1;
    return r;
  }
};
}  // namespace
void StepStackHandler::CreateArcCenteredInStepStack(double R, double arc_deg, int SupportFoot) {
  const double kPi = 3.14159265358979323846;
  const double StepMax = 0.10;
  const double OmegaTotal = arc_deg * kPi / 180.0;
  const int NumberOfStep = (int)floor(OmegaTotal * R / StepMax);
  const double LastStep = OmegaTotal * R - NumberOfStep * StepMax;
  const double OmegaStep = StepMax / R;
  const double LastOmegaStep = OmegaTotal - OmegaStep * NumberOfStep;
  if (SupportFoot * OmegaStep < 0.0) {             // the support foot must be the one that does not lead the motion
    push_rel(m_RelativeFootPositions, 0, -SupportFoot * 0.095, 0, m_SingleSupportTime, m_DoubleSupportTime);
    SupportFoot = -SupportFoot;
  }
  const double S = -SupportFoot * 0.095;
  const M3 Romegastep = M3::rigid(OmegaStep, 0, 0);
  const M3 MFNSF = M3::rigid(0, -R, S), MFSF = M3::rigid(0, -R, -S), Mtmp = M3::rigid(0, 0, 0.19);
  M3 MSupportFoot = MFSF;
  auto pair_of_steps = [&](const M3 &Romega, const M3 &turn, double dOmega) {
    const M3 RiR = (MSupportFoot * turn).rigid_inverse();
    const M3 FPos = RiR * (Romega * MFNSF);
    push_rel(m_RelativeFootPositions, FPos.a[0][2], FPos.a[1][2], dOmega * 180.0 / kPi, m_SingleSupportTime, m_DoubleSupportTime);
    MSupportFoot = Romega * MFNSF;
    push_rel(m_RelativeFootPositions, 0, SupportFoot * 0.19, 0, m_SingleSupportTime, m_DoubleSupportTime);
    MSupportFoot = MSupportFoot * Mtmp;
  };
  for (int i = 0; i < NumberOfStep; i++) pair_of_steps(M3::rigid((i + 1) * OmegaStep, 0, 0), Romegastep, OmegaStep);
  if (LastStep != 0.0)
    pair_of_steps(M3::rigid(LastOmegaStep + NumberOfStep * OmegaStep, 0, 0), M3::rigid(LastOmegaStep, 0, 0), LastOmegaStep);
  m_KeepLastCorrectSupportFoot = -SupportFoot;
}
void StepStackHandler::PushFrontAStepInTheStack(RelativeFootPosition &aRFP) { m_RelativeFootPositions.push_front(aRFP); }   // :865-868
void StepStackHandler::CallMethod(string &Method, istringstream &strm) {   // :929-1040, the commands restated here
  if (Method == ":singlesupporttime") strm >> m_SingleSupportTime;
  else if (Method == ":doublesupporttime") strm >> m_DoubleSupportTime;
  else if (Method == ":walkmode") strm >> m_WalkMode;
  else if (Method == ":supportfoot") { int f = -1; strm >> f; PrepareForSupportFoot(f); }
  else if (Method == ":lastsupport") FinishOnTheLastCorrectSupportFoot();
  else if (Method == ":arc") {
    double x = 0, y = 0, arc_deg = 0;
    int f = -1;
    strm >> x >> y >> arc_deg >> f;
    CreateArcInStepStack(x, y, 0.0, arc_deg, f);
  } else if (Method == ":arccentered") {
    double R = 0, arc_deg = 0;
    int f = -1;
    strm >> R >> arc_deg >> f;
    if (!(R != 0.0)) throw runtime_error(":arccentered needs a non-zero radius");
    CreateArcCenteredInStepStack(R, arc_deg, f);
  }
}
void StepStackHandler::CopyRelativeFootPosition(deque<RelativeFootPosition> &lRelativeFootPositions, bool PerformClean) {
  lRelativeFootPositions = m_RelativeFootPositions;
  if (PerformClean) m_RelativeFootPositions.clear();
}

// ---- FootConstraintsAsLinearSystem ----------------------------------------------------------------------------------------------
FootConstraintsAsLinearSystem::FootConstraintsAsLinearSystem(SimplePluginManager *aSPM, const HumanoidModel *aHS)
    : SimplePlugin(aSPM) {
  if (aHS == 0) throw runtime_error("FootConstraintsAsLinearSystem: a HumanoidModel is required");
  m_HS = *aHS;
}
FootConstraintsAsLinearSystem::~FootConstraintsAsLinearSystem() {}
void FootConstraintsAsLinearSystem::CallMethod(string &, istringstream &) {}   // FootConstraintsAsLinearSystem.cpp:541-545

// :258-539
int FootConstraintsAsLinearSystem::BuildLinearConstraintInequalities(deque<FootAbsolutePosition> &LeftFootAbsolutePositions,
                                                                     deque<FootAbsolutePosition> &RightFootAbsolutePositions,
                                                                     deque<LinearConstraintInequality_t *> &Queue,
                                                                     double ConstraintOnX, double ConstraintOnY) {
  if (LeftFootAbsolutePositions.size() != RightFootAbsolutePositions.size()) return -1;
  const size_t n = LeftFootAbsolutePositions.size();
  vector<double> time(n), lf(6 * n), rf(6 * n);
  vector<int> lty(n);
  for (size_t i = 0; i < n; i++) {
    const FootAbsolutePosition &l = LeftFootAbsolutePositions[i], &r = RightFootAbsolutePositions[i];
    time[i] = l.time;
    lty[i] = l.stepType;
    const double lv[6] = {l.x, l.y, l.z, l.theta, l.omega, l.omega2}, rv[6] = {r.x, r.y, r.z, r.theta, r.omega, r.omega2};
    for (int c = 0; c < 6; c++) { lf[6 * i + c] = lv[c]; rf[6 * i + c] = rv[c]; }
  }
  int cap = 64;
  vector<wg_zmp_polytope_t> polys;
  vector<double> ts, te;
  int k;
  for (;;) {
    polys.resize(cap); ts.resize(cap); te.resize(cap);
    k = wg_foot_constraints((int)n, time.data(), lf.data(), lty.data(), rf.data(), m_HS.soleWidth, m_HS.soleHeight,
                            ConstraintOnX, ConstraintOnY, cap, polys.data(), ts.data(), te.data());
    if (k < 0) return -1;
    if (k <= cap) break;
    cap = k;
  }
  for (int q = 0; q < k; q++) {
    LinearConstraintInequality_t *aLCI = new LinearConstraintInequality_t;
    const wg_zmp_polytope_t &P = polys[q];
    for (int j = 0; j < P.nrows; j++) {
      aLCI->A.push_back(P.A[j][0]); aLCI->A.push_back(P.A[j][1]);
      aLCI->B.push_back(P.B[j]);
      aLCI->SimilarConstraints.push_back(P.similar[j]);
    }
    aLCI->Center.push_back(P.centre[0]); aLCI->Center.push_back(P.centre[1]);
    aLCI->StartingTime = ts[q];
    aLCI->EndingTime = te[q];
    Queue.push_back(aLCI);
  }
  return 0;
}

namespace {

class PatternGeneratorInterfacePrivate : public virtual PatternGeneratorInterface, SimplePluginManager, SimplePlugin {
 public:
  PatternGeneratorInterfacePrivate(const HumanoidModel *aHDR)
      : PatternGeneratorInterface(aHDR), SimplePlugin(this), m_Model(*aHDR), m_Robot(0) { Construct(); }
  PatternGeneratorInterfacePrivate(CjrlHumanoidDynamicRobot *aHDR)
      : PatternGeneratorInterface(aHDR), SimplePlugin(this), m_Model(HumanoidModelFromRobot(aHDR)), m_Robot(aHDR) { Construct(); }
  void Construct() {
    // PatternGeneratorInterfacePrivate.cpp:181-215: the commands this object handles itself
    // the reference's fifteen (PatternGeneratorInterfacePrivate.cpp:186-201, in its order) ...
    string aMethodName[15] = {":LimitsFeasibility", ":ZMPShiftParameters", ":TimeDistributionParameters", ":stepseq", ":finish",
                              ":StartOnLineStepSequencing", ":StopOnLineStepSequencing", ":readfilefromkw", ":SetAlgoForZmpTrajectory",
                              ":SetAutoFirstStep", ":ChangeNextStep", ":samplingperiod", ":HerdtOnline", ":setVelReference",
                              ":setCoMPerturbationForce"};
    for (int i = 0; i < 15; i++)
      if (!SimplePlugin::RegisterMethod(aMethodName[i])) cerr << "Unable to register " << aMethodName[i] << endl;
    // ... the step-stack commands (StepStackHandler is a plugin of this manager in the reference, StepStackHandler.cpp:61-77) and
    // this build's switch for the golden file's legacy semantics
    string aStackMethod[8] = {":supportfoot", ":arc", ":arccentered", ":lastsupport", ":singlesupporttime", ":doublesupporttime",
                              ":walkmode", ":wg_legacy_golden"};
    for (int i = 0; i < 8; i++)
      if (!SimplePlugin::RegisterMethod(aStackMethod[i])) cerr << "Unable to register " << aStackMethod[i] << endl;
    // :62-79, :115-119
    m_AutoFirstStep = false;
    m_TimeDistrFactor.assign({2.0, 3.7, 1.0, 3.0});
    m_DeltaFeasibilityLimit = 0.0;
    m_ZMPShift.assign({0.02, 0.07, 0.02, 0.02});
    m_ZMPVRQP = new ZMPVelocityReferencedQP(this, "", &m_Model);
    m_ZMPD = new ZMPDiscretization(this, "", &m_Model);                                    // :223-226
    m_PC = new PreviewControl(this, OptimalControllerSolver::MODE_WITHOUT_INITIALPOS, true);   // :254
    m_NL = 0;
    m_SamplingPeriod = 0.005;
    m_InternalClock = 0.0;
    m_ShouldBeRunning = false;
    m_Running = false;
    m_Herdt = false;
    m_NbOfHitBottom = 0;
    // :147-174
    m_count = 0; m_dt = 0.005; m_AbsTheta = 0; m_AbsMotionTheta = 0;
    m_ZMPInitialPointSet = false;
    m_NewStep = false; m_NewStepX = m_NewStepY = m_NewTheta = 0.0;
    m_TSsupport = 0.78; m_TDsupport = 0.02;                 // :82-83 (":singlesupporttime", ":doublesupporttime")
    for (int i = 0; i < 4; i++) { m_AbsLinearVelocity[i] = 0.0; m_AbsLinearAcc[i] = 0.0; m_AbsAngularVelocity[i] = 0.0; }
    m_CurrentWaistState.reset();
  }
  ~PatternGeneratorInterfacePrivate() { delete m_ZMPVRQP; delete m_ZMPD; delete m_PC; }

  int ParseCmd(istringstream &strm) {                     // :1029-1039
    string aCmd;
    strm >> aCmd;
    if (SimplePluginManager::CallMethod(aCmd, strm)) {}
    return 0;
  }

  void CallMethod(string &aCmd, istringstream &strm) {    // :1055-1133 (the commands that exist on this path)
    if (aCmd == ":samplingperiod") {
      double sp;
      strm >> sp;
      m_SamplingPeriod = sp;
    } else if (aCmd == ":setVelReference") {
      m_ZMPVRQP->Reference(strm);
    } else if (aCmd == ":HerdtOnline") {
      m_InternalClock = 0.0;
      initOnlineHerdt();
      printf("Online \n");
    } else if (aCmd == ":setCoMPerturbationForce") {
      m_ZMPVRQP->setCoMPerturbationForce(strm);
    } else if (aCmd == ":SetAlgoForZmpTrajectory") {
      string ZMPTrajAlgo;
      strm >> ZMPTrajAlgo;
      if (ZMPTrajAlgo == "Herdt") { m_Herdt = true; cout << "Herdt" << endl; }
      else if (ZMPTrajAlgo == "Kajita") m_Herdt = false;             // the reference's default (ZMPCOM_KAJITA_2003)
      else { m_Herdt = false; cerr << "wg: the Herdt and Kajita (stage 1) generators are built (asked for " << ZMPTrajAlgo << ")" << endl; }
    } else if (aCmd == ":stepseq") {                                 // m_StepSequence, :562-571
      m_SSH.ReadStepSequenceAccordingToWalkMode(strm);
      FinishAndRealizeStepSequence();
    } else if (aCmd == ":finish") {                                  // m_FinishAndRealizeStepSequence
      FinishAndRealizeStepSequence();
    } else if (aCmd == ":ZMPShiftParameters") {                      // m_SetZMPShiftParameters, :415-446: reaches ZMPDiscretization
      ReadUpToFour(strm, m_ZMPShift);                                //   through SetZMPShift at the next :stepseq / :finish (:895)
    } else if (aCmd == ":TimeDistributionParameters") {              // m_SetTimeDistrParameters, :1518-1548: kept; its consumer is
      ReadUpToFour(strm, m_TimeDistrFactor);                         //   the stepping-over planner (:981), which is not on this path
    } else if (aCmd == ":LimitsFeasibility") {                       // m_SetLimitsFeasibility, :448-460: kept; consumer as above (:483)
      while (!strm.eof()) { if (!(strm >> m_DeltaFeasibilityLimit)) break; }   // a non-numeric token ends it (failbit never reaches eof)
    } else if (aCmd == ":SetAutoFirstStep") {                        // :1122-1131
      string lAutoFirstStep;
      strm >> lAutoFirstStep;
      if (lAutoFirstStep == "true") m_AutoFirstStep = true;
      else if (lAutoFirstStep == "false") m_AutoFirstStep = false;
    } else if (aCmd == ":StartOnLineStepSequencing") {               // :1088-1093; the method throws NotOnThisPath
      m_InternalClock = 0.0;
      ReadSequenceOfSteps(strm);
      StartOnLineStepSequencing();
    } else if (aCmd == ":StopOnLineStepSequencing") {                // :1094-1095
      StopOnLineStepSequencing();
    } else if (aCmd == ":ChangeNextStep") {                          // :1060-1064: parsed, then refused like every generator but
      double nt;                                                     //   Morisawa's refuses it (ChangeOnLineStep returns -1)
      ChangeOnLineStep(strm, nt);
    } else if (aCmd == ":readfilefromkw") {                          // m_ReadFileFromKineoWorks, :1581-1623: KineoWorks paths feed the
      cerr << "wg: :readfilefromkw ignored (GenerateMotionFromKineoWorks is not on the path of this build)" << endl;   // upper-body planner
    } else if (aCmd == ":supportfoot" || aCmd == ":arc" || aCmd == ":arccentered" || aCmd == ":lastsupport" ||
               aCmd == ":singlesupporttime" || aCmd == ":doublesupporttime" || aCmd == ":walkmode") {
      if (aCmd == ":singlesupporttime" || aCmd == ":doublesupporttime") {   // :1085-1094: kept for AddStepInStack too
        istringstream peek(strm.str());
        string skip;
        double v = 0.0;
        peek >> skip >> v;
        (aCmd == ":singlesupporttime" ? m_TSsupport : m_TDsupport) = v;
      }
      m_SSH.CallMethod(aCmd, strm);                                  // StepStackHandler is a plugin of this manager there
    } else if (aCmd == ":wg_legacy_golden") {
      int on = 0;
      strm >> on;
      m_ZMPVRQP->LegacyGoldenReplay(on != 0);
    }
  }

  // ComAndFootRealizationByGeometry::InitializationFoot, ComAndFootRealizationByGeometry.cpp:381-440: the foot frame is the
  // ankle frame moved by minus the ankle's position in the foot frame; its rotation is taken relative to the ankle's rotation
  // in the reference posture; the foot must be flat
  static void InitializationFoot(const CjrlFoot *aFoot, FootAbsolutePosition &InitFootPosition) {
    const CjrlJoint *AnkleJoint = aFoot->associatedAnkle();
    if (AnkleJoint == 0) throw runtime_error("EvaluateStartingState: a foot without an ankle joint");
    vector3d anklePosition;
    aFoot->getAnklePositionInLocalFrame(anklePosition);
    matrix4d lFootPose = AnkleJoint->currentTransformation();
    for (int i = 0; i < 3; i++) {                       // lFootPose * translation(-anklePosition)
      double t = lFootPose(i, 3);
      for (int k = 0; k < 3; k++) t += lFootPose(i, k) * -anklePosition[k];
      lFootPose(i, 3) = t;
    }
    InitFootPosition.x = lFootPose(0, 3); InitFootPosition.y = lFootPose(1, 3); InitFootPosition.z = lFootPose(2, 3);
    const matrix4d &initialRot = AnkleJoint->initialPosition();
    double invrot[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        invrot[i][j] = 0.0;
        for (int k = 0; k < 3; k++) invrot[i][j] += lFootPose(i, k) * initialRot(j, k);
      }
    if (invrot[2][1] != 0.0) throw runtime_error("EvaluateStartingState: the initial foot position is not flat");   // assert there
    InitFootPosition.omega = atan2(invrot[2][0], invrot[2][2]) * 180 / kPi;
    InitFootPosition.theta = atan2(-invrot[0][1], invrot[1][1]) * 180 / kPi;
  }

  // the strategy-level evaluation (CoMAndFootOnlyStrategy::EvaluateStartingState, CoMAndFootOnlyStrategy.cpp:127-155 ->
  // ComAndFootRealizationByGeometry::InitializationCoM, ComAndFootRealizationByGeometry.cpp:449-545), without the
  // ":comheight" command the public method adds
  void StrategyEvaluateStartingState(COMState &lStartingCOMState, vector3d &lStartingZMPPosition, vectorN &lStartingWaistPose,
                                     FootAbsolutePosition &InitLeftFootAbsPos, FootAbsolutePosition &InitRightFootAbsPos) {
    memset(&InitLeftFootAbsPos, 0, sizeof InitLeftFootAbsPos);
    memset(&InitRightFootAbsPos, 0, sizeof InitRightFootAbsPos);
    if (m_Robot == 0) {                                   // plain numbers: the start state is part of them
      lStartingCOMState.reset();
      lStartingCOMState.x[0] = m_Model.startCoM[0]; lStartingCOMState.y[0] = m_Model.startCoM[1];
      lStartingCOMState.z[0] = m_Model.startCoM[2]; lStartingCOMState.z[1] = lStartingCOMState.z[2] = 0.0;
      for (int i = 0; i < 3; i++) lStartingZMPPosition[i] = m_Model.startZMP[i];
      lStartingWaistPose.assign(6, 0.0);
      InitLeftFootAbsPos.x = m_Model.startLeftFoot[0]; InitLeftFootAbsPos.y = m_Model.startLeftFoot[1];
      InitLeftFootAbsPos.theta = m_Model.startLeftFoot[2];
      InitRightFootAbsPos.x = m_Model.startRightFoot[0]; InitRightFootAbsPos.y = m_Model.startRightFoot[1];
      InitRightFootAbsPos.theta = m_Model.startRightFoot[2];
      return;
    }
    // InitializationHumanoid :305-380: free flyer at the given waist pose (zero if none), joints after it
    const unsigned int nDof = m_Robot->numberDof();
    if (nDof < 6 + m_CurrentJointValues.size())
      throw runtime_error("EvaluateStartingState: more joint values than the robot has degrees of freedom");
    vectorN CurrentConfig(nDof, 0.0);
    if (lStartingWaistPose.size() >= 6) for (int i = 0; i < 6; i++) CurrentConfig[i] = lStartingWaistPose[i];
    else lStartingWaistPose.assign(6, 0.0);
    for (size_t i = 0; i < m_CurrentJointValues.size(); i++) CurrentConfig[6 + i] = m_CurrentJointValues[i];
    m_Robot->currentConfiguration(CurrentConfig);
    {
      string inProperty[2] = {"ComputeCoM", "ComputeZMP"};
      string inValue[2] = {"true", "false"};
      for (unsigned int i = 0; i < 2; i++) m_Robot->setProperty(inProperty[i], inValue[i]);
    }
    m_Robot->computeForwardKinematics();
    CurrentConfig = m_Robot->currentConfiguration();
    for (int i = 0; i < 6; i++) lStartingWaistPose[i] = CurrentConfig[i];
    CjrlFoot *RightFoot = m_Robot->rightFoot(), *LeftFoot = m_Robot->leftFoot();
    InitializationFoot(RightFoot, InitRightFootAbsPos);
    InitializationFoot(LeftFoot, InitLeftFootAbsPos);
    vector3d COGInitialAnkles(0.5 * (InitRightFootAbsPos.x + InitLeftFootAbsPos.x), 0.5 * (InitRightFootAbsPos.y + InitLeftFootAbsPos.y),
                              0.5 * (InitRightFootAbsPos.z + InitLeftFootAbsPos.z));
    lStartingWaistPose[2] -= InitRightFootAbsPos.z;
    vector3d lStartingCOMPosition = m_Robot->positionCenterOfMass();
    lStartingCOMPosition[2] -= InitRightFootAbsPos.z;
    InitLeftFootAbsPos.z = 0.0;
    InitRightFootAbsPos.z = 0.0;
    lStartingCOMState.x[0] = lStartingCOMPosition[0]; lStartingCOMState.y[0] = lStartingCOMPosition[1];
    lStartingCOMState.z[0] = lStartingCOMPosition[2];
    lStartingCOMState.yaw[0] = lStartingWaistPose[5]; lStartingCOMState.pitch[0] = lStartingWaistPose[4];
    lStartingCOMState.roll[0] = lStartingWaistPose[3];
    lStartingZMPPosition = COGInitialAnkles;
  }

  void EvaluateStartingState(COMState &lStartingCOMState, vector3d &lStartingZMPPosition, vectorN &lStartingWaistPose,
                             FootAbsolutePosition &InitLeftFootAbsPos, FootAbsolutePosition &InitRightFootAbsPos) {   // :588-617
    StrategyEvaluateStartingState(lStartingCOMState, lStartingZMPPosition, lStartingWaistPose, InitLeftFootAbsPos,
                                  InitRightFootAbsPos);
    ostringstream osscomheightcmd;
    osscomheightcmd.precision(17);
    osscomheightcmd << ":comheight " << lStartingCOMState.z[0];
    istringstream isscomheightcmd(osscomheightcmd.str());
    ParseCmd(isscomheightcmd);
  }

  void initOnlineHerdt() {                                // :517-560
    COMState lStartingCOMState;
    vector3d lStartingZMPPosition;
    vectorN lStartingWaistPose;
    FootAbsolutePosition InitLeftFootAbsPos, InitRightFootAbsPos;
    EvaluateStartingState(lStartingCOMState, lStartingZMPPosition, lStartingWaistPose, InitLeftFootAbsPos, InitRightFootAbsPos);
    deque<RelativeFootPosition> RelativeFootPositions;
    m_ZMPVRQP->SetCurrentTime(m_InternalClock);
    m_ZMPVRQP->InitOnLine(m_ZMPPositions, m_COMBuffer, m_LeftFootPositions, m_RightFootPositions, InitLeftFootAbsPos,
                          InitRightFootAbsPos, RelativeFootPositions, lStartingCOMState, lStartingZMPPosition.v);
    m_NbOfHitBottom = 0;
    m_ShouldBeRunning = true;
  }

  // :688-776 (AutoFirstStep is off on this path, :62)
  void CommonInitializationOfWalking(COMState &lStartingCOMState, vector3d &lStartingZMPPosition, vectorN &BodyAnglesIni,
                                     FootAbsolutePosition &InitLeftFootAbsPos, FootAbsolutePosition &InitRightFootAbsPos,
                                     deque<RelativeFootPosition> &lRelativeFootPositions, vector<double> &lCurrentJointValues,
                                     bool ClearStepStackHandler) {
    m_ZMPPositions.clear();
    m_LeftFootPositions.clear();
    m_RightFootPositions.clear();
    lCurrentJointValues = m_CurrentJointValues;
    BodyAnglesIni = vectorN(m_CurrentJointValues);
    m_SSH.CopyRelativeFootPosition(lRelativeFootPositions, ClearStepStackHandler);
    vectorN lStartingWaistPose;
    StrategyEvaluateStartingState(lStartingCOMState, lStartingZMPPosition, lStartingWaistPose, InitLeftFootAbsPos,
                                  InitRightFootAbsPos);
    if (m_AutoFirstStep && !lRelativeFootPositions.empty()) {        // :734-749
      AutomaticallyAddFirstStep(lRelativeFootPositions, InitLeftFootAbsPos, InitRightFootAbsPos, lStartingCOMState);
      if (!ClearStepStackHandler) m_SSH.PushFrontAStepInTheStack(lRelativeFootPositions[0]);
    }
    if (m_Robot) { string aProperty("ResetIteration"), aValue("any"); m_Robot->setProperty(aProperty, aValue); }
  }

  // :619-686: the step from the CoM's frame to the foot the first given step leaves on the ground, pushed in front of the
  // sequence (3 x 3 homogeneous transforms: inverse(CoM pose) * pose of that foot).  Angles as the reference takes them
  // there: radians, straight from the start state.
  void AutomaticallyAddFirstStep(deque<RelativeFootPosition> &lRelativeFootPositions, FootAbsolutePosition &InitLeftFootAbsPos,
                                 FootAbsolutePosition &InitRightFootAbsPos, COMState &lStartingCOMState) {
    const double cc = cos(lStartingCOMState.yaw[0]), sc = sin(lStartingCOMState.yaw[0]);
    const FootAbsolutePosition &F = lRelativeFootPositions[0].sy > 0 ? InitRightFootAbsPos : InitLeftFootAbsPos;
    const double cf = cos(F.theta), sf = sin(F.theta);
    // inverse of [cc -sc x; sc cc y; 0 0 1] times [cf -sf fx; sf cf fy; 0 0 1]
    const double dx = F.x - lStartingCOMState.x[0], dy = F.y - lStartingCOMState.y[0];
    const double m00 = cc * cf + sc * sf, m10 = -sc * cf + cc * sf;
    RelativeFootPosition aRFP;
    memset(&aRFP, 0, sizeof aRFP);
    aRFP.sx = cc * dx + sc * dy;
    aRFP.sy = -sc * dx + cc * dy;
    aRFP.theta = atan2(m10, m00);
    lRelativeFootPositions.push_front(aRFP);
  }
  static void ReadUpToFour(istringstream &strm, vector<double> &v) {   // the parsing loop of :415-446 / :1518-1548
    while (!strm.eof()) {
      for (int i = 0; i < 4; i++) { if (strm.eof() || !(strm >> v[i])) return; }   // also ends on a non-numeric token (failbit)
    }
  }

  // :881-1005 in Kajita mode, stage 1 only: the step stack -> ZMPDiscretization (CreateZMPReferences :1870-1882) -> the
  // first preview loop over the whole queue in one launch (the reference runs it NL samples ahead of the second, multi-body
  // stage inside DoubleStagePreviewControlStrategy; that second stage needs the robot model and is out of scope, so the
  // CoM handed out is the cart-table one)
  void FinishAndRealizeStepSequence() {
    COMState lStartingCOMState;
    vector3d lStartingZMPPositionV;
    vectorN BodyAnglesIni;
    FootAbsolutePosition InitLeftFootAbsPos, InitRightFootAbsPos;
    deque<RelativeFootPosition> lRelativeFootPositions;
    vector<double> lCurrentJointValues;
    m_ZMPD->SetZMPShift(m_ZMPShift);                                                                 // :895
    CommonInitializationOfWalking(lStartingCOMState, lStartingZMPPositionV, BodyAnglesIni, InitLeftFootAbsPos, InitRightFootAbsPos,
                                  lRelativeFootPositions, lCurrentJointValues, true);                  // :897-904
    if (m_ZMPInitialPointSet) lStartingZMPPositionV = m_ZMPInitialPoint;
    double *lStartingZMPPosition = lStartingZMPPositionV.v;
    m_COMBuffer.clear();
    m_ZMPD->SetCurrentTime(m_InternalClock);
    m_ZMPD->GetZMPDiscretization(m_ZMPPositions, m_COMBuffer, lRelativeFootPositions, m_LeftFootPositions,
                                 m_RightFootPositions, 0.0, lStartingCOMState, lStartingZMPPosition, InitLeftFootAbsPos,
                                 InitRightFootAbsPos);
    if (!m_PC->IsCoherent()) throw runtime_error("PatternGeneratorInterface: preview-control gains are not set (:samplingperiod, :previewcontroltime, :comheight)");
    m_NL = (unsigned)(m_PC->PreviewControlTime() / m_PC->SamplingPeriod());
    const size_t nq = m_ZMPPositions.size();
    if (nq < 2 * (size_t)m_NL) throw runtime_error("PatternGeneratorInterface: step sequence shorter than the preview windows");
    const int L = (int)(nq - m_NL + 1);
    vector<double> zx(nq), zy(nq), com((size_t)L * 6), st(8, 0.0);
    for (size_t i = 0; i < nq; i++) { zx[i] = m_ZMPPositions[i].px; zy[i] = m_ZMPPositions[i].py; }
    if (m_PC->RunBatch(1, L, zx.data(), zy.data(), st.data(), com.data(), 0, true) != 0) wg_throw("wg_preview_run_batch");
    for (size_t i = 0; i < nq; i++) {
      const size_t r = i < (size_t)L ? i : (size_t)L - 1;
      for (int k = 0; k < 3; k++) { m_COMBuffer[i].x[k] = com[6 * r + k]; m_COMBuffer[i].y[k] = com[6 * r + 3 + k]; }
    }
    m_NbOfHitBottom = 0;
    m_ShouldBeRunning = true;
  }

  // CoMAndFootOnlyStrategy::EndOfMotion with m_BufferSizeLimit = 0, CoMAndFootOnlyStrategy.cpp:157-188
  int EndOfMotion() {
    if (m_LeftFootPositions.size() > 0) { m_NbOfHitBottom = 0; return 1; }
    if (m_NbOfHitBottom == 0) { m_NbOfHitBottom++; return 0; }
    return -1;
  }

  // :1246-1514 (Herdt branch) + CoMAndFootOnlyStrategy::OneGlobalStepOfControl
  bool RunOneStepOfTheControlLoop(vectorN &CurrentConfiguration, vectorN &CurrentVelocity, vectorN &CurrentAcceleration,
                                  vectorN &ZMPTarget, COMState &finalCOMState, FootAbsolutePosition &LeftFootPosition,
                                  FootAbsolutePosition &RightFootPosition) {
    const bool r = OneStep(ZMPTarget, finalCOMState, LeftFootPosition, RightFootPosition);
    if (r) {
      // :1360-1375: the waist state is read from the caller's configuration (this strategy does not write it)
      m_count++;
      if (CurrentConfiguration.size() >= 6) {
        m_CurrentWaistState.x[0] = CurrentConfiguration[0]; m_CurrentWaistState.y[0] = CurrentConfiguration[1];
        m_CurrentWaistState.z[0] = CurrentConfiguration[2]; m_CurrentWaistState.roll[0] = CurrentConfiguration[3];
        m_CurrentWaistState.pitch[0] = CurrentConfiguration[4]; m_CurrentWaistState.yaw[0] = CurrentConfiguration[5];
      }
      if (CurrentVelocity.size() >= 3) {
        m_CurrentWaistState.x[1] = CurrentVelocity[0]; m_CurrentWaistState.y[1] = CurrentVelocity[1];
        m_CurrentWaistState.z[1] = CurrentVelocity[2];
      }
      (void)CurrentAcceleration;
    }
    // :1508-1510: the absolute motion frame moves on when a motion has finished
    UpdateAbsolutePosition(!r || !m_ShouldBeRunning);
    return r;
  }
  bool OneStep(vectorN &ZMPTarget, COMState &finalCOMState, FootAbsolutePosition &LeftFootPosition,
               FootAbsolutePosition &RightFootPosition) {
    m_InternalClock += m_SamplingPeriod;
    if (!m_Herdt && m_ShouldBeRunning && m_ZMPPositions.size() <= 2 * (size_t)m_NL) {
      m_ShouldBeRunning = false;   // DoubleStagePreviewControlStrategy::EndOfMotion: two preview windows stay on the queue
    }
    if ((!m_ShouldBeRunning) || (EndOfMotion() < 0)) {
      m_Running = false;
      return m_Running;
    }
    m_Running = true;
    if (m_Herdt) {
      m_ZMPVRQP->OnLine(m_InternalClock, m_ZMPPositions, m_COMBuffer, m_LeftFootPositions, m_RightFootPositions);
      m_Running = m_ZMPVRQP->Running() || m_ZMPVRQP->LegacyGoldenReplay();
    }
    if (m_LeftFootPositions.size() > 0) { LeftFootPosition = m_LeftFootPositions[0]; m_LeftFootPositions.pop_front(); }
    else return m_Running = false;
    if (m_RightFootPositions.size() > 0) { RightFootPosition = m_RightFootPositions[0]; m_RightFootPositions.pop_front(); }
    if (m_COMBuffer.size() > 0) { finalCOMState = m_COMBuffer[0]; m_COMBuffer.pop_front(); }
    if (m_ZMPPositions.size() > 0) {
      ZMPTarget.assign(3, 0.0);
      ZMPTarget[0] = m_ZMPPositions[0].px; ZMPTarget[1] = m_ZMPPositions[0].py; ZMPTarget[2] = 0;
      m_ZMPPositions.pop_front();
    }
    return m_Running;
  }
  bool RunOneStepOfTheControlLoop(vectorN &q, vectorN &dq, vectorN &ddq, vectorN &ZMPTarget, COMPosition &finalCOMPosition,
                                  FootAbsolutePosition &L, FootAbsolutePosition &R) {
    COMState aCOMState;
    m_Running = RunOneStepOfTheControlLoop(q, dq, ddq, ZMPTarget, aCOMState, L, R);
    finalCOMPosition = aCOMState;
    return m_Running;
  }
  bool RunOneStepOfTheControlLoop(vectorN &q, vectorN &dq, vectorN &ddq, vectorN &ZMPTarget) {
    FootAbsolutePosition L, R;
    COMState c;
    return RunOneStepOfTheControlLoop(q, dq, ddq, ZMPTarget, c, L, R);
  }
  bool RunOneStepOfTheControlLoop(FootAbsolutePosition &L, FootAbsolutePosition &R, ZMPPosition &ZMPRefPos,
                                  COMPosition &COMRefPos) {    // :1198-1229
    vectorN q, dq, ddq, ZMPTarget;
    COMState c;
    bool r = RunOneStepOfTheControlLoop(q, dq, ddq, ZMPTarget, c, L, R);
    if (ZMPTarget.size() >= 3) { ZMPRefPos.px = ZMPTarget[0]; ZMPRefPos.py = ZMPTarget[1]; ZMPRefPos.pz = ZMPTarget[2]; }
    COMRefPos = c;
    return r;
  }
  void SetCurrentJointValues(vectorN &v) { m_CurrentJointValues = v; }                     // :1549-1558
  void AddStepInStack(double dx, double dy, double theta) { m_SSH.AddStepInTheStack(dx, dy, theta, m_TSsupport, m_TDsupport); }   // :1906-1912
  int GetWalkMode() const { return m_SSH.GetWalkMode(); }                                  // :1576-1579
  void GetLegJointVelocity(vectorN &dqr, vectorN &dql) const { dqr.assign(6, 0.0); dql.assign(6, 0.0); }   // :1588-1600
  void ReadSequenceOfSteps(istringstream &strm) {                                          // :461-515
    if (m_SSH.GetWalkMode() != 0) throw NotOnThisPath("ReadSequenceOfSteps in walk mode " + to_string(m_SSH.GetWalkMode()));
    m_SSH.ReadStepSequenceAccordingToWalkMode(strm);
  }
  void StartOnLineStepSequencing() { throw NotOnThisPath("StartOnLineStepSequencing (on-line step stack)"); }
  void StopOnLineStepSequencing() {}                                                       // :876-879: a flag of the step stack
  void AddOnLineStep(double X, double Y, double Theta) { m_NewStep = true; m_NewStepX = X; m_NewStepY = Y; m_NewTheta = Theta; }   // :1625-1631
  int ChangeOnLineStep(double, FootAbsolutePosition &, double &) { return -1; }            // :1795-1816: Morisawa only
  void ChangeOnLineStep(istringstream &strm, double &newtime) {                            // :1041-1053
    FootAbsolutePosition aFAP;
    double ltime;
    strm >> ltime; strm >> aFAP.x; strm >> aFAP.y; strm >> aFAP.theta;
    ChangeOnLineStep(ltime, aFAP, newtime);
  }
  void setZMPInitialPoint(vector3d &lZMPInitialPoint) { m_ZMPInitialPoint = lZMPInitialPoint; m_ZMPInitialPointSet = true; }   // :1914-1918
  void getZMPInitialPoint(vector3d &lZMPInitialPoint) const { lZMPInitialPoint = m_ZMPInitialPoint; }

  // ---- odometry, :1632-1780 ----
  static void mul4(matrix4d &C, const matrix4d &A, const matrix4d &B) {
    matrix4d R;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double t = 0.0;
        for (int k = 0; k < 4; k++) t += A(i, k) * B(k, j);
        R(i, j) = t;
      }
    C = R;
  }
  void UpdateAbsolutePosition(bool UpdateAbsMotionOrNot) {
    const double thetarad = m_CurrentWaistState.yaw[0];
    const double c = cos(thetarad), s = sin(thetarad);
    m_WaistRelativePos = matrix4d();
    m_WaistRelativePos(0, 0) = c; m_WaistRelativePos(0, 1) = -s;
    m_WaistRelativePos(1, 0) = s; m_WaistRelativePos(1, 1) = c;
    const double RelativeLinearVelocity[4] = {m_CurrentWaistState.x[1], m_CurrentWaistState.y[1], m_CurrentWaistState.z[0], 1.0};
    const double RelativeLinearAcc[4] = {m_CurrentWaistState.x[2], m_CurrentWaistState.y[2], 0.0, 1.0};
    for (int i = 0; i < 4; i++) {
      double v = 0.0, a = 0.0;
      for (int k = 0; k < 4; k++) { v += m_MotionAbsOrientation(i, k) * RelativeLinearVelocity[k]; a += m_MotionAbsOrientation(i, k) * RelativeLinearAcc[k]; }
      m_AbsLinearVelocity[i] = v; m_AbsLinearAcc[i] = a;
    }
    m_WaistRelativePos(0, 3) = m_CurrentWaistState.x[0];
    m_WaistRelativePos(1, 3) = m_CurrentWaistState.y[0];
    m_WaistRelativePos(2, 3) = m_CurrentWaistState.z[0];
    mul4(m_WaistAbsPos, m_MotionAbsPos, m_WaistRelativePos);
    m_AbsAngularVelocity[0] = 0.0; m_AbsAngularVelocity[1] = 0.0;
    if (m_count != 0) m_AbsAngularVelocity[2] = (m_AbsMotionTheta + thetarad - m_AbsTheta) / m_dt;
    else m_AbsAngularVelocity[2] = 0.0;
    m_AbsAngularVelocity[3] = 1.0;
    m_AbsTheta = fmod(m_AbsMotionTheta + thetarad, 2 * kPi);
    if (UpdateAbsMotionOrNot) {
      m_MotionAbsPos = m_WaistAbsPos;
      m_MotionAbsPos(2, 3) = 0.0;                     // the position is supposed at the ground level
      m_AbsMotionTheta = m_AbsTheta;
    }
  }
  void getWaistPositionMatrix(matrix4d &lWaistAbsPos) const { lWaistAbsPos = m_WaistAbsPos; }
  void getWaistPositionAndOrientation(double aTQ[7], double &Orientation) const {
    aTQ[0] = m_WaistAbsPos(0, 3); aTQ[1] = m_WaistAbsPos(1, 3); aTQ[2] = m_WaistAbsPos(2, 3);
    aTQ[3] = 0; aTQ[4] = 0; aTQ[5] = sin(0.5 * m_AbsTheta); aTQ[6] = cos(0.5 * m_AbsTheta);   // a yaw-only quaternion (x y z w)
    Orientation = m_AbsTheta;
  }
  void setWaistPositionAndOrientation(double aTQ[7]) {
    m_WaistAbsPos(0, 3) = aTQ[0]; m_WaistAbsPos(1, 3) = aTQ[1]; m_WaistAbsPos(2, 3) = aTQ[2];
    const double _x = aTQ[3], _y = aTQ[4], _z = aTQ[5], _r = aTQ[6];
    const double x2 = _x * _x, y2 = _y * _y, z2 = _z * _z, r2 = _r * _r;
    m_WaistAbsPos(0, 0) = r2 + x2 - y2 - z2; m_WaistAbsPos(1, 1) = r2 - x2 + y2 - z2; m_WaistAbsPos(2, 2) = r2 - x2 - y2 + z2;
    const double xy = _x * _y, yz = _y * _z, zx = _z * _x, rx = _r * _x, ry = _r * _y, rz = _r * _z;
    m_WaistAbsPos(0, 1) = 2 * (xy - rz); m_WaistAbsPos(0, 2) = 2 * (zx + ry); m_WaistAbsPos(1, 0) = 2 * (xy + rz);
    m_WaistAbsPos(1, 2) = 2 * (yz - rx); m_WaistAbsPos(2, 0) = 2 * (zx - ry); m_WaistAbsPos(2, 1) = 2 * (yz + rx);
  }
  void getWaistVelocity(double &dx, double &dy, double &omega) const {
    dx = m_AbsLinearVelocity[0]; dy = m_AbsLinearVelocity[1]; omega = m_AbsAngularVelocity[2];
  }
  void setVelocityReference(double x, double y, double yaw) { m_ZMPVRQP->Reference(x, y, yaw); }
  void setCoMPerturbationForce(double x, double y) { m_ZMPVRQP->setCoMPerturbationForce(x, y); }

 private:
  HumanoidModel m_Model;
  CjrlHumanoidDynamicRobot *m_Robot;          // the caller's robot (abstract-robot factory) or 0 (plain numbers)
  // odometry (:1632-1780) and the small members of the reference the extra interface methods read or write
  COMState m_CurrentWaistState;
  matrix4d m_WaistRelativePos, m_WaistAbsPos, m_MotionAbsPos, m_MotionAbsOrientation;
  double m_AbsLinearVelocity[4], m_AbsLinearAcc[4], m_AbsAngularVelocity[4];
  double m_AbsTheta, m_AbsMotionTheta, m_dt;
  unsigned long m_count;
  vector3d m_ZMPInitialPoint;
  bool m_ZMPInitialPointSet, m_NewStep;
  double m_NewStepX, m_NewStepY, m_NewTheta, m_TSsupport, m_TDsupport;
  // :62-79, :115-119: what ":SetAutoFirstStep", ":TimeDistributionParameters", ":LimitsFeasibility", ":ZMPShiftParameters" set
  bool m_AutoFirstStep;
  vector<double> m_TimeDistrFactor, m_ZMPShift;
  double m_DeltaFeasibilityLimit;
  ZMPVelocityReferencedQP *m_ZMPVRQP;
  ZMPDiscretization *m_ZMPD;
  PreviewControl *m_PC;
  StepStackHandler m_SSH;
  unsigned m_NL;
  deque<ZMPPosition> m_ZMPPositions;
  deque<COMState> m_COMBuffer;
  deque<FootAbsolutePosition> m_LeftFootPositions, m_RightFootPositions;
  vector<double> m_CurrentJointValues;
  double m_SamplingPeriod, m_InternalClock;
  bool m_ShouldBeRunning, m_Running, m_Herdt;
  int m_NbOfHitBottom;
};

}  // namespace

PatternGeneratorInterface *patternGeneratorInterfaceFactory(const HumanoidModel *aHDR) {
  if (aHDR == 0) throw runtime_error("patternGeneratorInterfaceFactory: a HumanoidModel is required");
  return new PatternGeneratorInterfacePrivate(aHDR);
}
PatternGeneratorInterface *patternGeneratorInterfaceFactory(CjrlHumanoidDynamicRobot *aHDR) {   // patterngeneratorinterface.hh:306
  if (aHDR == 0) throw runtime_error("patternGeneratorInterfaceFactory: a robot is required");
  return new PatternGeneratorInterfacePrivate(aHDR);
}

}  // namespace PatternGeneratorJRL
