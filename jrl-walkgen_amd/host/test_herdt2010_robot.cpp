// test_herdt2010_robot.cpp -- the reference's TestHerdt2010 the way the reference's own harness runs it
// (tests/TestObject.cpp:138-260, 515-605): a CjrlHumanoidDynamicRobot is created, handed to
// patternGeneratorInterfaceFactory(CjrlHumanoidDynamicRobot *), the joint values of the start posture go in through
// SetCurrentJointValues, and ":HerdtOnline" makes the generator evaluate its starting state from the robot's forward
// kinematics (PatternGeneratorInterfacePrivate.cpp:573-617) -- no start state is supplied by hand.
//
// The robot here is a small kinematic model written against include/wg_abstract_robot.hh (the reference's harness loads
// jrl-dynamics' sample.wrl, which is not in this image): waist + two legs of [hip yaw, hip pitch, knee, ankle pitch], point
// masses on waist / thighs / shanks / feet, feet flat when pitch = (-a, 2a, -a).  Nothing in the generator knows that.
//
// Output: line 1 "start ..." = what EvaluateStartingState returned and the model numbers the generator read from the robot,
// then the 38-column trace of the EmergencyStop profile, then "end ..." = the odometry answers of the interface.
//   test_herdt2010_robot out.dat [knee_angle_deg] [margin_x margin_y]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/wg_walkgen.hh"

using namespace PatternGeneratorJRL;
using namespace std;

namespace {
const double kPi = 3.14159265358979323846;

matrix4d mul(const matrix4d &A, const matrix4d &B) {
  matrix4d C;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { double t = 0.0; for (int k = 0; k < 4; k++) t += A(i, k) * B(k, j); C(i, j) = t; }
  return C;
}
matrix4d rotz(double a) { matrix4d M; M(0, 0) = cos(a); M(0, 1) = -sin(a); M(1, 0) = sin(a); M(1, 1) = cos(a); return M; }
matrix4d roty(double a) { matrix4d M; M(0, 0) = cos(a); M(0, 2) = sin(a); M(2, 0) = -sin(a); M(2, 2) = cos(a); return M; }
matrix4d trans(double x, double y, double z) { matrix4d M; M(0, 3) = x; M(1, 3) = y; M(2, 3) = z; return M; }

struct MockJoint : public CjrlJoint {
  double lo, hi, vmax;
  matrix4d cur, init;
  MockJoint() : lo(0), hi(0), vmax(0) {}
  double lowerBound(unsigned int) const { return lo; }
  double upperBound(unsigned int) const { return hi; }
  double upperVelocityBound(unsigned int) const { return vmax; }
  const matrix4d &currentTransformation() const { return cur; }
  const matrix4d &initialPosition() const { return init; }
};
struct MockFoot : public CjrlFoot {
  MockJoint *ankle;
  double soleLength, soleWidth, ankleZ;
  const CjrlJoint *associatedAnkle() const { return ankle; }
  void getSoleSize(double &outLength, double &outWidth) const { outLength = soleLength; outWidth = soleWidth; }
  void getAnklePositionInLocalFrame(vector3d &out) const { out = vector3d(0.0, 0.0, ankleZ); }
};
// 6 free-flyer dofs + 2 x 4 leg joints
struct MockRobot : public CjrlHumanoidDynamicRobot {
  MockJoint waistJ, leg[2][4];           // [0] left, [1] right: hip yaw, hip pitch, knee, ankle pitch
  MockFoot foot[2];
  vectorN cfg;
  vector3d com;
  double thigh, shank, hipY, hipZ, comX;
  double mWaist, mThigh, mShank, mFoot;
  MockRobot() : cfg(14, 0.0), thigh(0.3), shank(0.3), hipY(0.09), hipZ(-0.05), comX(0.04), mWaist(40.0), mThigh(4.0), mShank(3.0), mFoot(1.0) {
    for (int s = 0; s < 2; s++) {
      foot[s].ankle = &leg[s][3]; foot[s].soleLength = 0.25; foot[s].soleWidth = 0.14; foot[s].ankleZ = 0.105;
      leg[s][0].lo = -35.0 / 180.0 * kPi; leg[s][0].hi = 40.0 / 180.0 * kPi; leg[s][0].vmax = 3.5;   // hip yaw
    }
    currentConfiguration(cfg);
    computeForwardKinematics();
    for (int s = 0; s < 2; s++) for (int j = 0; j < 4; j++) leg[s][j].init = leg[s][j].cur;   // reference posture: all zero
    waistJ.init = waistJ.cur;
  }
  double mass() const { return mWaist + 2 * (mThigh + mShank + mFoot); }
  CjrlFoot *leftFoot() const { return const_cast<MockFoot *>(&foot[0]); }
  CjrlFoot *rightFoot() const { return const_cast<MockFoot *>(&foot[1]); }
  CjrlJoint *waist() const { return const_cast<MockJoint *>(&waistJ); }
  std::vector<CjrlJoint *> jointsBetween(const CjrlJoint &a, const CjrlJoint &b) const {
    std::vector<CjrlJoint *> r;
    if (&a != &waistJ) return r;
    for (int s = 0; s < 2; s++)
      if (&b == &leg[s][3]) { r.push_back(const_cast<MockJoint *>(&waistJ)); for (int j = 0; j < 4; j++) r.push_back(const_cast<MockJoint *>(&leg[s][j])); }
    return r;
  }
  unsigned int numberDof() const { return 14; }
  std::vector<CjrlJoint *> getActuatedJoints() const {
    std::vector<CjrlJoint *> r;
    for (int s = 0; s < 2; s++) for (int j = 0; j < 4; j++) r.push_back(const_cast<MockJoint *>(&leg[s][j]));
    return r;
  }
  bool setProperty(std::string &, const std::string &) { return true; }
  bool currentConfiguration(const vectorN &c) { if (c.size() != 14) return false; cfg = c; return true; }
  const vectorN &currentConfiguration() const { return cfg; }
  bool computeForwardKinematics() {
    // free flyer: translation, then roll-pitch-yaw (only yaw is used here)
    waistJ.cur = mul(trans(cfg[0], cfg[1], cfg[2]), rotz(cfg[5]));
    double m = mWaist;
    double acc[3] = {mWaist * (waistJ.cur(0, 3) + waistJ.cur(0, 0) * comX), mWaist * (waistJ.cur(1, 3) + waistJ.cur(1, 0) * comX), mWaist * waistJ.cur(2, 3)};
    for (int s = 0; s < 2; s++) {
      const double *q = &cfg[6 + 4 * s];
      const double y = s == 0 ? hipY : -hipY;
      leg[s][0].cur = mul(mul(waistJ.cur, trans(0.0, y, hipZ)), rotz(q[0]));
      leg[s][1].cur = mul(leg[s][0].cur, roty(q[1]));
      leg[s][2].cur = mul(mul(leg[s][1].cur, trans(0.0, 0.0, -thigh)), roty(q[2]));
      leg[s][3].cur = mul(mul(leg[s][2].cur, trans(0.0, 0.0, -shank)), roty(q[3]));
      const matrix4d midThigh = mul(leg[s][1].cur, trans(0.0, 0.0, -0.5 * thigh)), midShank = mul(leg[s][2].cur, trans(0.0, 0.0, -0.5 * shank));
      for (int k = 0; k < 3; k++) acc[k] += mThigh * midThigh(k, 3) + mShank * midShank(k, 3) + mFoot * leg[s][3].cur(k, 3);
      m += mThigh + mShank + mFoot;
    }
    com = vector3d(acc[0] / m, acc[1] / m, acc[2] / m);
    return true;
  }
  const vector3d &positionCenterOfMass() const { return com; }
};

void cmd(PatternGeneratorInterface &aPGI, const char *c) { istringstream strm(c); aPGI.ParseCmd(strm); }
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s out.dat [knee_angle_deg] [margin_x margin_y]\n", argv[0]); return 2; }
  const double a = (argc > 2 ? atof(argv[2]) : 25.0) / 180.0 * kPi;
  try {
    CjrlHumanoidDynamicRobot *aHDR = new MockRobot();
    PatternGeneratorInterface *aPGI = patternGeneratorInterfaceFactory(aHDR);      // patterngeneratorinterface.hh:306
    // TestObject::CreateAndInitializeHumanoidRobot, tests/TestObject.cpp:176-260
    const unsigned int lNbDofs = aHDR->numberDof();
    const unsigned int lNbActuatedJoints = (unsigned int)aHDR->getActuatedJoints().size();
    MAL_VECTOR_DIM(InitialPosition, double, lNbActuatedJoints);
    for (int s = 0; s < 2; s++) { InitialPosition(4 * s + 0) = 0.0; InitialPosition(4 * s + 1) = -a; InitialPosition(4 * s + 2) = 2 * a; InitialPosition(4 * s + 3) = -a; }
    aPGI->SetCurrentJointValues(InitialPosition);
    cmd(*aPGI, ":walkmode 0");
    MAL_VECTOR_DIM(CurrentConfiguration, double, lNbDofs);
    MAL_VECTOR_DIM(CurrentVelocity, double, lNbDofs);
    MAL_VECTOR_DIM(CurrentAcceleration, double, lNbDofs);
    MAL_VECTOR_DIM(ZMPTarget, double, 3);
    // CommonInitialization, tests/CommonTools.cpp:53-75 (first nine commands)
    const char *lBuffer[9] = {":comheight 0.8078", ":samplingperiod 0.005", ":previewcontroltime 1.6", ":omega 0.0",
                              ":stepheight 0.07", ":singlesupporttime 0.78", ":doublesupporttime 0.02", ":armparameters 0.5",
                              ":LimitsFeasibility 0.0"};
    for (int i = 0; i < 9; i++) cmd(*aPGI, lBuffer[i]);
    if (argc > 4) { string c = string(":setfeetconstraint XY ") + argv[3] + " " + argv[4]; cmd(*aPGI, c.c_str()); }
    // what the generator will start from: asked for directly first (the public method), then used by ":HerdtOnline"
    COMState c0;
    MAL_S3_VECTOR_TYPE(double) z0;
    MAL_VECTOR_TYPE(double) w0;
    FootAbsolutePosition L0, R0;
    aPGI->EvaluateStartingState(c0, z0, w0, L0, R0);
    const HumanoidModel hm = HumanoidModelFromRobot(aHDR);
    ofstream aof(argv[1]);
    aof.precision(17);
    aof << "start " << c0.x[0] << " " << c0.y[0] << " " << c0.z[0] << " " << z0[0] << " " << z0[1] << " " << z0[2] << " " << L0.x << " "
        << L0.y << " " << L0.theta << " " << R0.x << " " << R0.y << " " << R0.theta << " " << w0[2] << " " << hm.mass << " "
        << hm.soleWidth << " " << hm.soleHeight << " " << hm.leftHipYawLower << " " << hm.leftHipYawUpper << " "
        << hm.hipYawVelocityMax << " " << aPGI->GetWalkMode() << endl;
    // startEmergencyStop, TestHerdt2010.cpp:88-116
    cmd(*aPGI, ":SetAlgoForZmpTrajectory Herdt");
    cmd(*aPGI, ":singlesupporttime 0.7");
    cmd(*aPGI, ":doublesupporttime 0.1");
    cmd(*aPGI, ":HerdtOnline 0.2 0.0 0.2");
    cmd(*aPGI, ":numberstepsbeforestop 2");
    aof.precision(12);
    aof.setf(ios::scientific, ios::floatfield);
    COMState c;
    FootAbsolutePosition L, R;
    unsigned long it = 0;
    bool ok = true;
    while (ok && it < 6000) {
      it++;
      ok = aPGI->RunOneStepOfTheControlLoop(CurrentConfiguration, CurrentVelocity, CurrentAcceleration, ZMPTarget, c, L, R);
      if (ok) {
        aof << it * 0.005 << " " << c.x[0] << " " << c.y[0] << " " << c.z[0] << " " << c.yaw[0] << " " << c.x[1] << " " << c.y[1]
            << " " << c.z[1] << " " << ZMPTarget[0] << " " << ZMPTarget[1] << " ";
        const FootAbsolutePosition *F[2] = {&L, &R};
        for (int f = 0; f < 2; f++)
          aof << F[f]->x << " " << F[f]->y << " " << F[f]->z << " " << F[f]->dx << " " << F[f]->dy << " " << F[f]->dz << " "
              << F[f]->ddx << " " << F[f]->ddy << " " << F[f]->ddz << " " << F[f]->theta << " " << F[f]->omega << " "
              << F[f]->omega2 << " ";
        aof << ZMPTarget[0] << " " << ZMPTarget[1] << " " << 0.0 << " " << 0.0 << endl;
        // a caller's whole-body stage would put the realised waist pose here; the cart-table CoM stands in for it
        CurrentConfiguration(0) = c.x[0]; CurrentConfiguration(1) = c.y[0]; CurrentConfiguration(2) = c.z[0];
        CurrentConfiguration(5) = c.yaw[0];
        CurrentVelocity(0) = c.x[1]; CurrentVelocity(1) = c.y[1];
      }
      if (it == 5 * 200) cmd(*aPGI, ":setVelReference  0.0 0.0 0.4");
      if (it == 10 * 200) cmd(*aPGI, ":setVelReference  0.2 0.0 -0.2");
      if (it == (unsigned long)(15.2 * 200)) aPGI->setVelocityReference(0.0, 0.0, 0.0);
      if (it == (unsigned long)(20.8 * 200)) { cmd(*aPGI, ":setVelReference  0.0 0.0 0.0"); cmd(*aPGI, ":stoppg"); }
    }
    // the rest of the interface: odometry, step stack, the methods that belong to other generators
    double TQ[7], Orientation = 0.0, dx = 0.0, dy = 0.0, omega = 0.0;
    aPGI->getWaistPositionAndOrientation(TQ, Orientation);
    aPGI->getWaistVelocity(dx, dy, omega);
    MAL_S4x4_MATRIX(W, double);
    aPGI->getWaistPositionMatrix(W);
    MAL_VECTOR_DIM(dqr, double, 6);
    MAL_VECTOR_DIM(dql, double, 6);
    aPGI->GetLegJointVelocity(dqr, dql);
    aPGI->AddStepInStack(0.1, -0.19, 5.0);
    MAL_S3_VECTOR(zi, double);
    zi[0] = 0.01; zi[1] = 0.02; zi[2] = 0.0;
    aPGI->setZMPInitialPoint(zi);
    MAL_S3_VECTOR(zo, double);
    aPGI->getZMPInitialPoint(zo);
    double newtime = 0.0;
    const int chg = aPGI->ChangeOnLineStep(0.1, L, newtime);
    int notOnPath = 0;
    try { aPGI->StartOnLineStepSequencing(); } catch (const NotOnThisPath &) { notOnPath = 1; }
    aof.unsetf(ios::floatfield);
    aof.precision(17);
    aof << "end " << TQ[0] << " " << TQ[1] << " " << TQ[2] << " " << TQ[5] << " " << TQ[6] << " " << Orientation << " " << W(0, 3) << " "
        << W(1, 3) << " " << dx << " " << dy << " " << omega << " " << dqr.size() << " " << zo[0] + zo[1] << " " << chg << " "
        << notOnPath << " " << c.x[0] << " " << c.y[0] << " " << c.yaw[0] << endl;
    aof.close();
    delete aPGI;
    delete aHDR;                                                   // the caller owns the robot (tests/TestObject.cpp:117-129)
    printf("rows %lu\n", it - 1);
  } catch (const std::exception &e) {
    fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  return 0;
}
