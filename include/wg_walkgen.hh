// wg_walkgen.hh -- host C++ facade over the C ABI (include/wg_mpc.h): the reference's PatternGeneratorInterface /
// SimplePlugin API for the Herdt-2010 path, same class names, method names, argument meaning and error behaviour, so a
// caller of jrl-walkgen can switch library for THIS path.  Built as lib/libwg_walkgen.so (links libwg_mpc.so).
//
//   reference interface                                                        here
//   include/jrl/walkgen/pgtypes.hh:51-154                                      COMPosition, COMState, ZMPPosition, FootAbsolutePosition,
//                                                                              RelativeFootPosition
//   src/SimplePlugin.hh:46-72, src/SimplePluginManager.hh:46-93                SimplePlugin, SimplePluginManager
//   src/ZMPRefTrajectoryGeneration/ZMPRefTrajectoryGeneration.hh               ZMPRefTrajectoryGeneration (on-line part)
//   src/ZMPRefTrajectoryGeneration/ZMPVelocityReferencedQP.hh:59-131           ZMPVelocityReferencedQP
//   src/PreviewControl/OptimalControllerSolver.hh:137-221, PreviewControl.hh:53-187   OptimalControllerSolver, PreviewControl (Kajita stage 1)
//   include/jrl/walkgen/patterngeneratorinterface.hh:72-306                    PatternGeneratorInterface: EVERY pure virtual of the
//                                                                              reference, in the reference's order (same vtable
//                                                                              layout), + both factories.  Methods that belong to
//                                                                              generators outside this path (on-line step
//                                                                              sequencing, Morisawa's ChangeOnLineStep) are
//                                                                              declared and answer as documented at each one.
//   abstract-robot-dynamics (CjrlHumanoidDynamicRobot, CjrlFoot, CjrlJoint)    include/wg_abstract_robot.hh: the methods this path
//                                                                              calls, same names and meaning
//
// Differences forced by the image (documented, not hidden):
//   * abstract-robot-dynamics and jrl-mal are absent.  wg_abstract_robot.hh declares the part of their interface the path
//     uses (robot: the ~14 numbers of SURVEY.md 8(b) + forward kinematics of the start posture; MAL_VECTOR = vectorN,
//     a std::vector<double> with operator(), MAL_S3_VECTOR = vector3d, MAL_S4x4_MATRIX = matrix4d).
//   * A robot can also be given as the plain struct HumanoidModel (the numbers themselves, start state supplied): fleets
//     and tests that have no kinematic model.
//   * There is no CPU fall-back: construction throws std::runtime_error when the HIP device or library is unusable.
#ifndef WG_WALKGEN_HH
#define WG_WALKGEN_HH

#include <cstring>
#include <deque>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include <stdexcept>

#include "wg_abstract_robot.hh"
#include "wg_mpc.h"

namespace PatternGeneratorJRL {

// thrown by the interface methods that belong to generators outside this path (documented at each declaration)
struct NotOnThisPath : public std::logic_error {
  explicit NotOnThisPath(const std::string &what) : std::logic_error(what + ": not on the Herdt-2010 / Kajita stage-1 path of this build") {}
};

// ---- pgtypes.hh ---------------------------------------------------------------------------------------------------
struct COMState_s;
struct COMPosition_s {
  double x[3], y[3];
  double z[3];
  double yaw, pitch, roll;
  COMPosition_s &operator=(const COMState_s &aCS);
};
typedef COMPosition_s COMPosition;
struct COMState_s {
  double x[3], y[3], z[3];
  double yaw[3], pitch[3], roll[3];
  COMState_s &operator=(const COMPosition_s &aCS);
  void reset();
  COMState_s();
};
typedef COMState_s COMState;
struct RelativeFootPosition_s {
  double sx, sy, theta;
  double SStime, DStime;
  int stepType;
  double DeviationHipHeight;
};
typedef RelativeFootPosition_s RelativeFootPosition;
struct ZMPPosition_s {
  double px, py, pz;
  double theta;
  double time;
  int stepType;
};
typedef ZMPPosition_s ZMPPosition;
struct FootAbsolutePosition_t {
  double x, y, z, theta, omega, omega2;
  double dx, dy, dz, dtheta, domega, domega2;
  double ddx, ddy, ddz, ddtheta, ddomega, ddomega2;
  double time;
  int stepType;
};
typedef FootAbsolutePosition_t FootAbsolutePosition;

// ---- what the path reads from the robot model -----------------------------------------------------------------------
struct HumanoidModel {
  double mass;                         // mass()                         ZMPVelocityReferencedQP.cpp:68, 97
  double soleWidth, soleHeight;        // leftFoot()->getSoleSize(w,h)   relative-feet-inequalities.cpp:158-172
  double anklePosition[3];             // getAnklePositionInLocalFrame   rigid-body-system.cpp:38, 176 (kept, unused by the tick)
  bool hasHipYawLimits;                // jointsBetween(waist, ankle)[1] bounds, OrientationsPreview.cpp:42-68
  double leftHipYawLower, leftHipYawUpper, rightHipYawLower, rightHipYawUpper, hipYawVelocityMax;
  // EvaluateStartingState outputs (PatternGeneratorInterfacePrivate.cpp:588-617) -- supplied, no kinematics here
  double startCoM[3];
  double startZMP[3];
  double startLeftFoot[3], startRightFoot[3];   // x, y, theta (degrees)
  // jrl-dynamics' sample robot in TestHerdt2010's starting posture (constants identified from the reference's golden
  // file)
  static HumanoidModel sampleRobot();
};
// The numbers the generators read from an abstract robot, through exactly the calls the reference makes:
// mass() (ZMPVelocityReferencedQP.cpp:68), leftFoot()->getSoleSize (relative-feet-inequalities.cpp:158-178: the LEFT foot's
// size is used for both feet), getAnklePositionInLocalFrame (rigid-body-system.cpp:38), the hip-yaw bounds of
// jointsBetween(waist, ankle)[1] (OrientationsPreview.cpp:42-68).  The start state stays zero: EvaluateStartingState
// computes it from the robot's forward kinematics.
HumanoidModel HumanoidModelFromRobot(CjrlHumanoidDynamicRobot *aHDR);

// ---- SimplePlugin / SimplePluginManager -------------------------------------------------------------------------------
class SimplePlugin;
class SimplePluginManager {
 public:
  struct ltstr {
    bool operator()(const std::string s1, const std::string s2) const { return strcmp(s1.c_str(), s2.c_str()) < 0; }
  };

 protected:
  std::multimap<std::string, SimplePlugin *, ltstr> m_SimplePlugins;

 public:
  inline SimplePluginManager() {}
  virtual ~SimplePluginManager();
  const std::multimap<std::string, SimplePlugin *, ltstr> &getSimplePlugins() const { return m_SimplePlugins; }
  bool RegisterMethod(std::string &MethodName, SimplePlugin *aSP);
  void UnregisterPlugin(SimplePlugin *aSP);
  bool CallMethod(std::string &MethodName, std::istringstream &istrm);
  void Print();
};

class SimplePlugin {
 private:
  SimplePluginManager *m_SimplePluginManager;
  friend class SimplePluginManager;

 public:
  inline SimplePlugin(SimplePluginManager *lSPM) : m_SimplePluginManager(lSPM) {}
  virtual ~SimplePlugin();
  bool RegisterMethod(std::string &MethodName);
  virtual void CallMethod(std::string &Method, std::istringstream &astrm) = 0;
  SimplePluginManager *getSimplePluginManager() const { return m_SimplePluginManager; }
};

// ---- ZMPRefTrajectoryGeneration (on-line part) --------------------------------------------------------------------------
class ZMPRefTrajectoryGeneration : public SimplePlugin {
 protected:
  double m_Tsingle, m_Tdble, m_SamplingPeriod, m_ModulationSupportCoefficient, m_Omega, m_PreviewControlTime,
      m_StepHeight, m_CurrentTime, m_ComHeight;
  bool m_OnLineMode;

 public:
  ZMPRefTrajectoryGeneration(SimplePluginManager *lSPM);
  virtual ~ZMPRefTrajectoryGeneration() {}
  void SetTSingleSupport(double v) { m_Tsingle = v; }
  double GetTSingleSupport() const { return m_Tsingle; }
  void SetTDoubleSupport(double v) { m_Tdble = v; }
  double GetTDoubleSupport() const { return m_Tdble; }
  void SetSamplingPeriod(double v) { m_SamplingPeriod = v; }
  double GetSamplingPeriod() const { return m_SamplingPeriod; }
  void SetCurrentTime(double v) { m_CurrentTime = v; }
  double GetCurrentTime() const { return m_CurrentTime; }
  bool GetOnLineMode();
  virtual int InitOnLine(std::deque<ZMPPosition> &FinalZMPPositions, std::deque<COMState> &COMStates,
                         std::deque<FootAbsolutePosition> &FinalLeftFootAbsolutePositions,
                         std::deque<FootAbsolutePosition> &FinalRightFootAbsolutePositions,
                         FootAbsolutePosition &InitLeftFootAbsolutePosition,
                         FootAbsolutePosition &InitRightFootAbsolutePosition,
                         std::deque<RelativeFootPosition> &RelativeFootPositions, COMState &lStartingCOMState,
                         double lStartingZMPPosition[3]) = 0;
  virtual void OnLine(double time, std::deque<ZMPPosition> &FinalZMPPositions, std::deque<COMState> &COMStates,
                      std::deque<FootAbsolutePosition> &FinalLeftFootAbsolutePositions,
                      std::deque<FootAbsolutePosition> &FinalRightFootAbsolutePositions) = 0;
  virtual void CallMethod(std::string &Method, std::istringstream &strm);
};

// ---- ZMPVelocityReferencedQP ----------------------------------------------------------------------------------------------
// solution_t (privatepgtypes.hh:330-377), the fields a caller of Solution() can still read
struct solution_t {
  int NbVariables, NbConstraints;   // n, m of the last QP
  int Fail;                          // QL ifail
  int NbIterations, NbActiveConstraints;
  double JerkX, JerkY;               // Solution_vec[0], Solution_vec[N]
  void reset();
};

class ZMPVelocityReferencedQP : public ZMPRefTrajectoryGeneration {
 public:
  ZMPVelocityReferencedQP(SimplePluginManager *SPM, std::string DataFile, const HumanoidModel *aHS = 0);
  ~ZMPVelocityReferencedQP();
  void CallMethod(std::string &Method, std::istringstream &strm);
  int InitOnLine(std::deque<ZMPPosition> &FinalZMPPositions, std::deque<COMState> &CoMStates,
                 std::deque<FootAbsolutePosition> &FinalLeftFootTraj_deq,
                 std::deque<FootAbsolutePosition> &FinalRightFootTraj_deq,
                 FootAbsolutePosition &InitLeftFootAbsolutePosition, FootAbsolutePosition &InitRightFootAbsolutePosition,
                 std::deque<RelativeFootPosition> &RelativeFootPositions, COMState &lStartingCOMState,
                 double lStartingZMPPosition[3]);
  void OnLine(double time, std::deque<ZMPPosition> &FinalZMPPositions, std::deque<COMState> &CoMStates,
              std::deque<FootAbsolutePosition> &FinalLeftFootTraj_deq,
              std::deque<FootAbsolutePosition> &FinalRightFootTraj_deq);
  void Reference(std::istringstream &strm) { strm >> State_.vref[0]; strm >> State_.vref[1]; strm >> State_.vref[2]; }
  inline void Reference(double dx, double dy, double dyaw) { State_.vref[0] = dx; State_.vref[1] = dy; State_.vref[2] = dyaw; }
  inline bool Running() { return Running_; }
  inline void EndingPhase(bool EndingPhase) { State_.ending_phase = EndingPhase ? 1 : 0; }
  void setCoMPerturbationForce(double x, double y);
  void setCoMPerturbationForce(std::istringstream &strm);
  solution_t &Solution() { return Solution_; }
  inline const int &QP_N(void) const { return Model_.N; }
  // the flat state / model behind the facade (fleet callers hand these to wg_mpc_tick_batch_dev themselves)
  const wg_model_t &Model() const { return Model_; }
  wg_gait_state_t &State() { return State_; }
  // Replays the revision that recorded TestHerdt2010EmergencyStopTestFGPI.datref: initial support
  // Y = 0.1, no stop-centring branch, Running() stays true until the queues drain.  Call before InitOnLine.
  void LegacyGoldenReplay(bool on);
  bool LegacyGoldenReplay() const { return Legacy_; }
  // ":setfeetconstraint XY mx my" (RelativeFeetInequalities::CallMethod, relative-feet-inequalities.cpp:322-342): security
  // margins of the ZMP polygon; the device model follows
  void SetFeetConstraint(double SecurityMarginX, double SecurityMarginY);
  // QPProblem::dump(const char *) / dump(double Time) (qp-problem.cpp:656-675): the problem of the NEXT tick -- Q, D, DU, DS, XL,
  // XU and the solver parameters in the reference's text format -- assembled on the device from the current state
  // (wg_mpc_assemble_batch).  OnLine writes "<dir>/Problem_<time>.dat" for a tick whose solve failed (Solution().Fail > 0,
  // ZMPVelocityReferencedQP.cpp:399-402) when the environment has WG_DUMP_FAILED_QP (= a directory, or 1 for /tmp), for every
  // tick with WG_DUMP_EVERY_QP (the Herdt QP keeps jerk and foot placement free, so a failing one is hard to come by).
  void dumpProblem(const char *FileName);
  void dumpProblem(double Time);

 private:
  void dumpState(const wg_gait_state_t &state, const char *FileName);
  ZMPVelocityReferencedQP(const ZMPVelocityReferencedQP &);              // owns a device context: not copyable
  ZMPVelocityReferencedQP &operator=(const ZMPVelocityReferencedQP &);
  wg_ctx_t *Ctx_;    // this object's device-side model tables and workspaces (the reference keeps them per object too)
  wg_model_t Model_;
  // the gait's flat state and the tick's outputs live in host-mapped memory (wg_host_alloc): OnLine ticks through
  // wg_mpc_tick_pinned -- no staging copies, no device synchronisation ("one robot")
  void *Pinned_;
  wg_gait_state_t &State_;
  wg_tick_out_t *Out_;
  solution_t Solution_;
  bool Running_, Legacy_;
  int NbStepsSSDS_;
  double RobotMass_, PerturbationAcceleration_[6];
  bool PerturbationOccured_;
};

// ---- OptimalControllerSolver / PreviewControl (Kajita stage 1) -----------------------------------------------------------
// src/PreviewControl/OptimalControllerSolver.hh:137-221 over wg_riccati_solve; matrices are row-major std::vector<double>
class OptimalControllerSolver {
 public:
  static const unsigned int MODE_WITHOUT_INITIALPOS = 1;
  static const unsigned int MODE_WITH_INITIALPOS = 0;
  OptimalControllerSolver(const std::vector<double> &A, const std::vector<double> &b, const std::vector<double> &c, double Q,
                          double R, unsigned int Nl);
  void ComputeWeights(unsigned int Mode);
  void GetF(std::vector<double> &LF) { LF = m_F; }
  void GetK(std::vector<double> &LK) { LK = m_K; }

 protected:
  std::vector<double> m_A, m_b, m_c, m_K, m_F;
  double m_Q, m_R;
  unsigned int m_Nl;
};

// src/PreviewControl/PreviewControl.hh:53-187.  x and y (MAL_MATRIX 3x1 in the reference) are std::vector<double>(3).
// Every OneIterationOfPreview* call runs on the GPU through wg_preview_run_batch (B = 1, L = 1); callers with many gaits
// or many steps use RunBatch, which maps to one launch.
class PreviewControl : public SimplePlugin {
 public:
  PreviewControl(SimplePluginManager *lSPM, unsigned int defaultMode = OptimalControllerSolver::MODE_WITH_INITIALPOS,
                 bool computeWeightsAutomatically = false);
  ~PreviewControl();
  void ReadPrecomputedFile(std::string aFileName);
  int OneIterationOfPreview(std::vector<double> &x, std::vector<double> &y, double &sxzmp, double &syzmp,
                            std::deque<ZMPPosition> &ZMPPositions, unsigned int lindex, double &zmpx2, double &zmpy2,
                            bool Simulation);
  int OneIterationOfPreview1D(std::vector<double> &x, double &sxzmp, std::deque<double> &ZMPPositions, unsigned int lindex,
                              double &zmpx2, bool Simulation);
  int OneIterationOfPreview1D(std::vector<double> &x, double &sxzmp, std::vector<double> &ZMPPositions, unsigned int lindex,
                              double &zmpx2, bool Simulation);
  // L steps of B gaits in one launch; arguments as wg_preview_run_batch (include/wg_mpc.h)
  int RunBatch(int B, int L, const double *zmp_x, const double *zmp_y, double *state, double *com, double *zmp2,
               bool Simulation);
  double SamplingPeriod() const { return m_SamplingPeriod; }
  double PreviewControlTime() const { return m_PreviewControlTime; }
  double GetHeightOfCoM() const { return m_Zc; }
  void SetSamplingPeriod(double lSamplingPeriod);
  void SetPreviewControlTime(double lPreviewControlTime);
  void SetHeightOfCoM(double lZc);
  bool IsCoherent() { return m_Coherent; }
  void ComputeOptimalWeights(unsigned int mode);
  void print();
  virtual void CallMethod(std::string &Method, std::istringstream &astrm);
  // gains as held (m_Kx, m_Ks, m_F)
  const double *Kx() const { return m_Kx; }
  double Ks() const { return m_Ks; }
  const std::vector<double> &F() const { return m_F; }

 private:
  void Upload();   // hands the current gains to the device side (wg_preview_configure) when they changed
  double m_Kx[3], m_Ks;
  std::vector<double> m_F;
  double m_PreviewControlTime, m_SamplingPeriod, m_Zc;
  unsigned int m_SizeOfPreviewWindow;
  bool m_Coherent, m_AutoComputeWeights, m_Uploaded;
  wg_ctx_t *m_Ctx;   // this object's device-side gains (created with the first Upload)
  unsigned int m_DefaultWeightComputationMode;
};

// ---- PatternGeneratorInterface ---------------------------------------------------------------------------------------------
// ZMPDiscretization (src/ZMPRefTrajectoryGeneration/ZMPDiscretization.hh): the ZMP reference and feet trajectories of a
// step sequence, Kajita mode.  Same entry points as the reference; every sample comes from wg_zmpdisc_batch (GPU).  The
// on-line calls (OnLineAddFoot after InitOnLine) re-run the sequence known so far and append the new samples: the
// producer is deterministic, so this equals continuing from the reference's member state.
class ZMPDiscretization : public ZMPRefTrajectoryGeneration {
 public:
  ZMPDiscretization(SimplePluginManager *lSPM, std::string DataFile = "", const HumanoidModel *aHS = 0);
  ~ZMPDiscretization();
  void GetZMPDiscretization(std::deque<ZMPPosition> &ZMPPositions, std::deque<COMState> &COMStates,
                            std::deque<RelativeFootPosition> &RelativeFootPositions,
                            std::deque<FootAbsolutePosition> &LeftFootAbsolutePositions,
                            std::deque<FootAbsolutePosition> &RightFootAbsolutePositions, double Xmax,
                            COMState &lStartingCOMState, double lStartingZMPPosition[3],
                            FootAbsolutePosition &InitLeftFootAbsolutePosition,
                            FootAbsolutePosition &InitRightFootAbsolutePosition);
  int InitOnLine(std::deque<ZMPPosition> &FinalZMPPositions, std::deque<COMState> &COMStates,
                 std::deque<FootAbsolutePosition> &FinalLeftFootAbsolutePositions,
                 std::deque<FootAbsolutePosition> &FinalRightFootAbsolutePositions,
                 FootAbsolutePosition &InitLeftFootAbsolutePosition, FootAbsolutePosition &InitRightFootAbsolutePosition,
                 std::deque<RelativeFootPosition> &RelativeFootPositions, COMState &lStartingCOMState,
                 double lStartingZMPPosition[3]);
  void OnLine(double time, std::deque<ZMPPosition> &FinalZMPPositions, std::deque<COMState> &COMStates,
              std::deque<FootAbsolutePosition> &FinalLeftFootAbsolutePositions,
              std::deque<FootAbsolutePosition> &FinalRightFootAbsolutePositions);
  void OnLineAddFoot(RelativeFootPosition &NewRelativeFootPosition, std::deque<ZMPPosition> &FinalZMPPositions,
                     std::deque<COMState> &COMStates, std::deque<FootAbsolutePosition> &FinalLeftFootAbsolutePositions,
                     std::deque<FootAbsolutePosition> &FinalRightFootAbsolutePositions, bool EndSequence);
  void EndPhaseOfTheWalking(std::deque<ZMPPosition> &ZMPPositions, std::deque<COMState> &FinalCOMStates,
                            std::deque<FootAbsolutePosition> &LeftFootAbsolutePositions,
                            std::deque<FootAbsolutePosition> &RightFootAbsolutePositions);
  int ReturnOptimalTimeToRegenerateAStep();
  void SetZMPShift(std::vector<double> &ZMPShift);
  void CallMethod(std::string &Method, std::istringstream &strm);
  const wg_zmpdisc_model_t &Model();

 private:
  // runs the whole sequence on the GPU and appends samples [from, to) (to < 0: up to the end phase / with it)
  void Produce(bool with_end, size_t from, std::deque<ZMPPosition> &Z, std::deque<COMState> &Cs,
               std::deque<FootAbsolutePosition> &L, std::deque<FootAbsolutePosition> &R);
  wg_zmpdisc_model_t Model_;
  std::vector<wg_rel_step_t> Steps_;
  double InitFeet_[6];
  double StartTime_;
  size_t Produced_;                 // samples handed out so far (without an end phase)
};

// StepStackHandler (src/StepStackHandler.hh), the part ":stepseq" needs: walk mode 0
class StepStackHandler {
 public:
  StepStackHandler() : m_SingleSupportTime(0.78), m_DoubleSupportTime(0.02), m_WalkMode(0) {}
  void SetSingleTimeSupport(double v) { m_SingleSupportTime = v; }
  void SetDoubleTimeSupport(double v) { m_DoubleSupportTime = v; }
  int GetWalkMode() const { return m_WalkMode; }
  void ReadStepSequenceAccordingToWalkMode(std::istringstream &strm);    // StepStackHandler.cpp:128-175
  void AddStepInTheStack(double sx, double sy, double theta, double sstime, double dstime);   // :850-863
  void CopyRelativeFootPosition(std::deque<RelativeFootPosition> &lRelativeFootPositions, bool PerformClean);
  // ":supportfoot", ":arc", ":lastsupport" (StepStackHandler.cpp:754-764, 299-457, 872-882; CallMethod :929-1040)
  void PrepareForSupportFoot(int SupportFoot);
  void CreateArcInStepStack(double x, double y, double R, double arc_deg, int SupportFoot);
  // ":arccentered R arc_deg support_foot" (StepStackHandler.cpp:459-752, CallMethod :1011-1037)
  void CreateArcCenteredInStepStack(double R, double arc_deg, int SupportFoot);
  void PushFrontAStepInTheStack(RelativeFootPosition &aRFP);                                   // :865-868
  void FinishOnTheLastCorrectSupportFoot();
  void CallMethod(std::string &Method, std::istringstream &strm);

 private:
  std::deque<RelativeFootPosition> m_RelativeFootPositions;
  double m_SingleSupportTime, m_DoubleSupportTime;
  int m_WalkMode, m_KeepLastCorrectSupportFoot = 1;
};

// LinearConstraintInequality_t (pgtypes.hh:168-177): A is rows x 2 (row-major), B rows x 1
struct LinearConstraintInequality_s {
  std::vector<double> A, B, Center;
  std::vector<int> SimilarConstraints;
  double StartingTime, EndingTime;
};
typedef LinearConstraintInequality_s LinearConstraintInequality_t;

// FootConstraintsAsLinearSystem (src/Mathematics/FootConstraintsAsLinearSystem.hh) over wg_foot_constraints
class FootConstraintsAsLinearSystem : public SimplePlugin {
 public:
  FootConstraintsAsLinearSystem(SimplePluginManager *aSPM, const HumanoidModel *aHS);
  ~FootConstraintsAsLinearSystem();
  int BuildLinearConstraintInequalities(std::deque<FootAbsolutePosition> &LeftFootAbsolutePositions,
                                        std::deque<FootAbsolutePosition> &RightFootAbsolutePositions,
                                        std::deque<LinearConstraintInequality_t *> &QueueOfLConstraintInequalities,
                                        double ConstraintOnX, double ConstraintOnY);
  void CallMethod(std::string &Method, std::istringstream &strm);

 private:
  HumanoidModel m_HS;
};

// include/jrl/walkgen/patterngeneratorinterface.hh:55-306 -- every pure virtual, in the reference's order
class PatternGeneratorInterface {
 public:
  PatternGeneratorInterface(CjrlHumanoidDynamicRobot *) {}
  PatternGeneratorInterface(const HumanoidModel *) {}
  virtual ~PatternGeneratorInterface() {}
  // :72  step stack (StepStackHandler::AddStepInTheStack)
  virtual void AddStepInStack(double dx, double dy, double theta) = 0;
  // :92  start state + copy of the step stack (PatternGeneratorInterfacePrivate.cpp:688-776; AutoFirstStep is off)
  virtual void CommonInitializationOfWalking(COMState &lStartingCOMState, MAL_S3_VECTOR_TYPE(double) & lStartingZMPPosition,
                                             MAL_VECTOR(&, double) BodyAnglesIni, FootAbsolutePosition &InitLeftFootAbsPos,
                                             FootAbsolutePosition &InitRightFootAbsPos,
                                             std::deque<RelativeFootPosition> &lRelativeFootPositions,
                                             std::vector<double> &lCurrentJointValues, bool ClearStepStackHandler) = 0;
  // :115-176  the control loop.  On this path (CoMAndFootOnlyStrategy) the configuration vectors are not written; their
  // first six entries, when present, are read as the waist state for the odometry below (:1366-1375)
  virtual bool RunOneStepOfTheControlLoop(MAL_VECTOR_TYPE(double) & CurrentConfiguration, MAL_VECTOR_TYPE(double) & CurrentVelocity,
                                          MAL_VECTOR_TYPE(double) & CurrentAcceleration, MAL_VECTOR_TYPE(double) & ZMPTarget) = 0;
  virtual bool RunOneStepOfTheControlLoop(MAL_VECTOR_TYPE(double) & CurrentConfiguration, MAL_VECTOR_TYPE(double) & CurrentVelocity,
                                          MAL_VECTOR_TYPE(double) & CurrentAcceleration, MAL_VECTOR_TYPE(double) & ZMPTarget,
                                          COMPosition &COMPosition, FootAbsolutePosition &LeftFootPosition,
                                          FootAbsolutePosition &RightFootPosition) = 0;
  virtual bool RunOneStepOfTheControlLoop(MAL_VECTOR_TYPE(double) & CurrentConfiguration, MAL_VECTOR_TYPE(double) & CurrentVelocity,
                                          MAL_VECTOR_TYPE(double) & CurrentAcceleration, MAL_VECTOR_TYPE(double) & ZMPTarget,
                                          COMState &COMState, FootAbsolutePosition &LeftFootPosition,
                                          FootAbsolutePosition &RightFootPosition) = 0;
  virtual bool RunOneStepOfTheControlLoop(FootAbsolutePosition &LeftFootPosition, FootAbsolutePosition &RightFootPosition,
                                          ZMPPosition &ZMPRefPos, COMPosition &COMRefPos) = 0;
  // :184
  virtual void SetCurrentJointValues(MAL_VECTOR(&lCurrentJointValues, double)) = 0;
  // :187
  virtual int GetWalkMode() const = 0;
  // :190  leg joint velocities come from the whole-body stage (out of scope): six zeros each, the reference's initial values
  virtual void GetLegJointVelocity(MAL_VECTOR(&dqr, double), MAL_VECTOR(&dql, double)) const = 0;
  // :194  walk mode 0 (":stepseq" format); other walk modes throw NotOnThisPath
  virtual void ReadSequenceOfSteps(std::istringstream &strm) = 0;
  // :200-206  on-line step sequencing drives the Kajita / Morisawa generators from the step stack while walking: throws
  // NotOnThisPath (Start), no effect (Stop, Add: they only set flags Start would read)
  virtual void StartOnLineStepSequencing() = 0;
  virtual void StopOnLineStepSequencing() = 0;
  virtual void AddOnLineStep(double X, double Y, double Theta) = 0;
  // :224-234  Morisawa-2007 only in the reference; any other generator answers -1 / does nothing there too (:1800-1816)
  virtual int ChangeOnLineStep(double Time, FootAbsolutePosition &aFootAbsolutePosition, double &newtime) = 0;
  virtual void ChangeOnLineStep(std::istringstream &strm, double &newtime) = 0;
  // :241-258  odometry (PatternGeneratorInterfacePrivate.cpp:1632-1780)
  virtual void UpdateAbsolutePosition(bool UpdateAbsMotionOrNot) = 0;
  virtual void getWaistPositionAndOrientation(double TQ[7], double &Orientation) const = 0;
  virtual void setWaistPositionAndOrientation(double TQ[7]) = 0;
  virtual void getWaistVelocity(double &dx, double &dy, double &omega) const = 0;
  virtual void getWaistPositionMatrix(MAL_S4x4_MATRIX(&lWaistAbsPos, double)) const = 0;
  // :264-267
  virtual void setZMPInitialPoint(MAL_S3_VECTOR(&, double) lZMPInitialPoint) = 0;
  virtual void getZMPInitialPoint(MAL_S3_VECTOR(&, double) lZMPInitialPoint) const = 0;
  // :274
  virtual int ParseCmd(std::istringstream &strm) = 0;
  // :283  from the robot's forward kinematics of the current joint values (abstract robot), or the supplied start state
  // (HumanoidModel); then ":comheight <CoM z>" like the reference (PatternGeneratorInterfacePrivate.cpp:588-617)
  virtual void EvaluateStartingState(COMState &lStartingCOMState, MAL_S3_VECTOR_TYPE(double) & lStartingZMPPosition,
                                     MAL_VECTOR_TYPE(double) & lStartingWaistPose, FootAbsolutePosition &InitLeftFootAbsPos,
                                     FootAbsolutePosition &InitRightFootAbsPos) = 0;
  // :298, :305
  virtual void setVelocityReference(double x, double y, double yaw) = 0;
  virtual void setCoMPerturbationForce(double x, double y) = 0;
};

// patterngeneratorinterface.hh:306; the caller owns the robot and the returned object
PatternGeneratorInterface *patternGeneratorInterfaceFactory(CjrlHumanoidDynamicRobot *aHDR);
// the same generator on the plain numbers (start state supplied in the struct)
PatternGeneratorInterface *patternGeneratorInterfaceFactory(const HumanoidModel *);

}  // namespace PatternGeneratorJRL
#endif
