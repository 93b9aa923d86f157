// test_herdt2010.cpp -- the reference's TestHerdt2010 "EmergencyStop" profile (tests/TestHerdt2010.cpp:88-116, 128-200,
// 260-265, event loop tests/TestObject.cpp:515-605, row layout :344-385) driven through this build's
// PatternGeneratorInterface.  Writes the 38-column trace to argv[1]; `--legacy` replays the revision that recorded the
// reference's golden file (see ZMPVelocityReferencedQP::LegacyGoldenReplay).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/wg_walkgen.hh"

using namespace PatternGeneratorJRL;
using namespace std;

static void cmd(PatternGeneratorInterface &aPGI, const char *c) {
  istringstream strm(c);
  aPGI.ParseCmd(strm);
}

int main(int argc, char **argv) {
  const char *path = 0;
  bool legacy = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--legacy")) legacy = true;
    else path = argv[i];
  }
  if (!path) { fprintf(stderr, "usage: %s [--legacy] out.dat\n", argv[0]); return 2; }
  try {
    HumanoidModel robot = HumanoidModel::sampleRobot();
    PatternGeneratorInterface *aPGI = patternGeneratorInterfaceFactory(&robot);
    // CommonInitialization, tests/CommonTools.cpp:53-75 (first nine commands)
    const char *lBuffer[9] = {":comheight 0.8078", ":samplingperiod 0.005", ":previewcontroltime 1.6", ":omega 0.0",
                              ":stepheight 0.07", ":singlesupporttime 0.78", ":doublesupporttime 0.02", ":armparameters 0.5",
                              ":LimitsFeasibility 0.0"};
    for (int i = 0; i < 9; i++) cmd(*aPGI, lBuffer[i]);
    if (legacy) cmd(*aPGI, ":wg_legacy_golden 1");
    // startEmergencyStop, TestHerdt2010.cpp:88-116
    cmd(*aPGI, ":SetAlgoForZmpTrajectory Herdt");
    cmd(*aPGI, ":singlesupporttime 0.7");
    cmd(*aPGI, ":doublesupporttime 0.1");
    cmd(*aPGI, ":HerdtOnline 0.2 0.0 0.2");
    cmd(*aPGI, ":numberstepsbeforestop 2");

    ofstream aof(path);
    aof.precision(12);
    aof.setf(ios::scientific, ios::floatfield);
    vector<double> q, dq, ddq, ZMPTarget(3, 0.0);
    COMState c;
    FootAbsolutePosition L, R;
    unsigned long it = 0;
    bool ok = true;
    while (ok && it < 6000) {
      it++;
      ok = aPGI->RunOneStepOfTheControlLoop(q, dq, ddq, ZMPTarget, c, L, R);
      if (ok) {
        aof << it * 0.005 << " " << c.x[0] << " " << c.y[0] << " " << c.z[0] << " " << c.yaw[0] << " " << c.x[1] << " " << c.y[1]
            << " " << c.z[1] << " " << ZMPTarget[0] << " " << ZMPTarget[1] << " ";
        const FootAbsolutePosition *F[2] = {&L, &R};
        for (int f = 0; f < 2; f++)
          aof << F[f]->x << " " << F[f]->y << " " << F[f]->z << " " << F[f]->dx << " " << F[f]->dy << " " << F[f]->dz << " "
              << F[f]->ddx << " " << F[f]->ddy << " " << F[f]->ddz << " " << F[f]->theta << " " << F[f]->omega << " "
              << F[f]->omega2 << " ";
        aof << ZMPTarget[0] << " " << ZMPTarget[1] << " " << 0.0 << " " << 0.0 << endl;
      }
      // generateEvent, TestHerdt2010.cpp:231-265
      if (it == 5 * 200) cmd(*aPGI, ":setVelReference  0.0 0.0 0.4");
      if (it == 10 * 200) cmd(*aPGI, ":setVelReference  0.2 0.0 -0.2");
      if (it == (unsigned long)(15.2 * 200)) cmd(*aPGI, ":setVelReference  0.0 0.0 0.0");
      if (it == (unsigned long)(20.8 * 200)) { cmd(*aPGI, ":setVelReference  0.0 0.0 0.0"); cmd(*aPGI, ":stoppg"); }
    }
    aof.close();
    delete aPGI;
    printf("rows %lu\n", it - 1);
  } catch (const std::exception &e) {
    fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  return 0;
}
