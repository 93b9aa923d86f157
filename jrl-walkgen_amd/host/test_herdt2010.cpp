// test_herdt2010.cpp -- the reference's TestHerdt2010 driven through this build's PatternGeneratorInterface: the
// "EmergencyStop" profile (tests/TestHerdt2010.cpp:88-116, 128-200, 260-265) by default, the "OnLine" profile (:64-91,
// 231-244: twelve events over 110 s; its golden file is not in the reference tree) with `--online`; event loop
// tests/TestObject.cpp:515-605, row layout :344-385.  Writes the 38-column trace to the path given; `--legacy` replays
// the revision that recorded the reference's golden file (see ZMPVelocityReferencedQP::LegacyGoldenReplay).
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
  bool legacy = false, online = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--legacy")) legacy = true;
    else if (!strcmp(argv[i], "--online")) online = true;
    else path = argv[i];
  }
  if (!path) { fprintf(stderr, "usage: %s [--legacy] [--online] out.dat\n", argv[0]); return 2; }
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
    cmd(*aPGI, online ? ":HerdtOnline 0.2 0.0 0.0" : ":HerdtOnline 0.2 0.0 0.2");
    cmd(*aPGI, ":numberstepsbeforestop 2");

    ofstream aof(path);
    aof.precision(12);
    aof.setf(ios::scientific, ios::floatfield);
    vectorN q, dq, ddq, ZMPTarget(3, 0.0);
    COMState c;
    FootAbsolutePosition L, R;
    unsigned long it = 0;
    bool ok = true;
    const unsigned long max_it = online ? 24000 : 6000;
    while (ok && it < max_it) {
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
      if (online) {
        static const struct { unsigned long time; const char *c; } ev[12] = {
            {5 * 200, ":setVelReference  0.2 0.0 0.0"},    {10 * 200, ":setVelReference  0.0 0.2 0.0"},
            {25 * 200, ":setVelReference  0.0 0.0 -10."},  {35 * 200, ":setVelReference  0.2 0.0 0.0"},
            {45 * 200, ":setVelReference  0.0 0.0 10.0"},  {55 * 200, ":setVelReference  0.2 0.0 0.0"},
            {65 * 200, ":setVelReference  0.0 0.0 -10."},  {75 * 200, ":setVelReference  0.2 0.0 0.0"},
            {85 * 200, ":setVelReference  0.2 0.0 6.0832"}, {95 * 200, ":setVelReference  0.2 0.0 -6.0832"},
            {105 * 200, ":setVelReference 0.0 0.0 0.0"},   {110 * 200, ":setVelReference  0.0 0.0 0.0"}};
        for (int e = 0; e < 12; e++)
          if (it == ev[e].time) cmd(*aPGI, ev[e].c);
        if (it == 110 * 200) cmd(*aPGI, ":stoppg");
        continue;
      }
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
