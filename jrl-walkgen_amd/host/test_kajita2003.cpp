// test_kajita2003.cpp -- the reference's TestKajita2003 (tests/TestKajita2003.cpp, tests/TestObject.cpp:515-605) re-hosted on
// the PatternGeneratorInterface of include/wg_walkgen.hh, Kajita mode, stage 1: CommonInitialization's commands
// (tests/CommonTools.cpp:57-70), then one of the test's profiles, then RunOneStepOfTheControlLoop until it says stop, one
// row of the reference's 38-column trace (tests/TestObject.cpp:344-385) per call.  Feet and ZMP-reference columns are the
// reference's (its golden files pin them); the CoM columns hold the FIRST preview stage's cart-table CoM (the reference
// prints the CoM after the second, multi-body stage, which needs the robot model), ZMPTarget is left in the world frame
// (columns 9-10 = 35-36) and the waist columns 37-38 are zero.
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/wg_walkgen.hh"

using namespace PatternGeneratorJRL;

static void cmd(PatternGeneratorInterface &pgi, const std::string &c) {
  std::istringstream s(c);
  pgi.ParseCmd(s);
}

int main(int argc, char **argv) {
  const std::string profile = argc > 1 ? argv[1] : "StraightWalking";
  const std::string out = argc > 2 ? argv[2] : ("TestKajita2003" + profile + "TestFGPI.dat");
  try {
    HumanoidModel robot = HumanoidModel::sampleRobot();
    robot.startLeftFoot[0] = 0.0094903; robot.startLeftFoot[1] = 0.095; robot.startLeftFoot[2] = 0.0;     // HRP-2 half-sitting,
    robot.startRightFoot[0] = 0.0094903; robot.startRightFoot[1] = -0.095; robot.startRightFoot[2] = 0.0;  // read off the goldens
    PatternGeneratorInterface *pgi = patternGeneratorInterfaceFactory(&robot);
    const char *common[] = {":comheight 0.8078", ":samplingperiod 0.005", ":previewcontroltime 1.6", ":omega 0.0", ":stepheight 0.07",
                            ":singlesupporttime 0.78", ":doublesupporttime 0.02", ":armparameters 0.5", ":LimitsFeasibility 0.0"};
    for (const char *c : common) cmd(*pgi, c);
    if (profile == "StraightWalking") {                       // tests/TestKajita2003.cpp:95-123
      cmd(*pgi, ":SetAlgoForZmpTrajectory Kajita");
      cmd(*pgi, ":stepseq 0.0 -0.105 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 "
                "0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.0 0.21 0.0");
    } else if (profile == "PbFlorentSeq1") {                  // :125-153
      cmd(*pgi, ":SetAlgoForZmpTrajectory Kajita");
      cmd(*pgi, ":stepseq 0 0.1 0 -0.0398822 -0.232351 4.6646 -0.0261703 0.199677 4.6646 -0.0471999 -0.256672 4.6646 "
                "-0.0305785 0.200634 4.6646 -0.0507024 -0.245393 4.6646 -0.0339626 0.197227 4.6646 -0.0527259 -0.228579 4.6646 "
                "-0.0362332 0.199282 4.6646 -0.0540087 -0.21638 4.6646 -0.0373302 0.196611 4.6646 -0.0536928 -0.199019 4.6646 "
                "-0.0372245 0.204021 4.6646 -0.0529848 -0.196642 4.6646 -0.0355124 0.2163 4.6646 -0.000858977 -0.204807 0.0767924 0 0.2 0");
    } else if (profile == "Circle") {                         // TurningOnTheCircle, :68-93
      cmd(*pgi, ":supportfoot 1");
      cmd(*pgi, ":arc 0.0 0.75 30.0 -1");
      cmd(*pgi, ":lastsupport");
      cmd(*pgi, ":finish");
    } else
      throw std::runtime_error("unknown profile " + profile);

    FILE *f = fopen(out.c_str(), "w");
    if (!f) throw std::runtime_error("cannot open " + out);
    vectorN q, dq, ddq, zmp;
    COMState com;
    FootAbsolutePosition lf, rf;
    unsigned long n = 0;
    while (pgi->RunOneStepOfTheControlLoop(q, dq, ddq, zmp, com, lf, rf)) {
      n++;
      const double row[38] = {n * 0.005, com.x[0], com.y[0], com.z[0], com.yaw[0], com.x[1], com.y[1], com.z[1], zmp[0], zmp[1],
                              lf.x, lf.y, lf.z, lf.dx, lf.dy, lf.dz, lf.ddx, lf.ddy, lf.ddz, lf.theta, lf.omega, lf.omega2,
                              rf.x, rf.y, rf.z, rf.dx, rf.dy, rf.dz, rf.ddx, rf.ddy, rf.ddz, rf.theta, rf.omega, rf.omega2,
                              zmp[0], zmp[1], 0.0, 0.0};
      for (int c = 0; c < 38; c++) fprintf(f, "%.10e ", row[c]);
      fprintf(f, "\n");
      if (n > 100000) throw std::runtime_error("the control loop does not end");
    }
    fclose(f);
    printf("TestKajita2003 %s (stage 1): %lu control steps, final CoM (%.6f, %.6f)\n", profile.c_str(), n, com.x[0], com.y[0]);
    delete pgi;
  } catch (std::exception &e) {
    std::cerr << "FAILED: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
