// test_kajita_preview.cpp -- BASELINE config[0] plumbing: the step sequence of the reference's
// TestKajita2003 StraightWalking (tests/TestKajita2003.cpp:96-124) through stage 1 only -- the ZMP reference queue and feet
// of ZMPDiscretization, then PreviewControl (both include/wg_walkgen.hh) -- on the GPU.  The first steps go through
// OneIterationOfPreview one call at a time (the reference's call pattern, ZMPPreviewControlWithMultiBodyZMP.cpp), the
// whole run through RunBatch; both must agree bit for bit.  Writes "t com_x com_y zmp_x zmp_y zmpref_x zmpref_y lf_x lf_y lf_z rf_x rf_y rf_z lf_theta rf_theta"; a second argument "Circle" runs TurningOnTheCircle instead.
#include <cstdio>
#include <cstring>
#include <deque>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "../../include/wg_walkgen.hh"

using namespace PatternGeneratorJRL;

int main(int argc, char **argv) {
  const char *out = argc > 1 ? argv[1] : "TestKajita2003StraightWalkingStage1.dat";
  const bool circle = argc > 2 && std::string(argv[2]) == "Circle";   // default: StraightWalking
  try {
    SimplePluginManager spm;
    PreviewControl pc(&spm, OptimalControllerSolver::MODE_WITHOUT_INITIALPOS, false);
    { std::string m(":samplingperiod"); std::istringstream s("0.005"); spm.CallMethod(m, s); }
    { std::string m(":previewcontroltime"); std::istringstream s("1.6"); spm.CallMethod(m, s); }
    { std::string m(":comheight"); std::istringstream s("0.814"); spm.CallMethod(m, s); }
    pc.ComputeOptimalWeights(OptimalControllerSolver::MODE_WITHOUT_INITIALPOS);
    if (!pc.IsCoherent()) throw std::runtime_error("gains not coherent");
    const double T = pc.SamplingPeriod();
    const unsigned nl = (unsigned)(pc.PreviewControlTime() / T);

    // CommonInitialization (tests/CommonTools.cpp:57-65) and StraightWalking's ":stepseq" (tests/TestKajita2003.cpp:104-120)
    // through StepStackHandler and ZMPDiscretization::GetZMPDiscretization, as PatternGeneratorInterfacePrivate does for
    // ":stepseq" in Kajita mode (m_StepSequence :562-571, CreateZMPReferences :1870-1882)
    HumanoidModel robot = HumanoidModel::sampleRobot();
    ZMPDiscretization zmpd(&spm, "", &robot);
    const char *cmds[] = {":omega 0.0", ":stepheight 0.07", ":singlesupporttime 0.78", ":doublesupporttime 0.02"};
    for (const char *c : cmds) { std::istringstream s(c); std::string m; s >> m; spm.CallMethod(m, s); }
    StepStackHandler ssh;
    ssh.SetSingleTimeSupport(0.78);
    ssh.SetDoubleTimeSupport(0.02);
    if (circle) {                      // TurningOnTheCircle, tests/TestKajita2003.cpp:68-93 (":finish" realises the sequence)
      const char *seq[] = {":supportfoot 1", ":arc 0.0 0.75 30.0 -1", ":lastsupport"};
      for (const char *c : seq) { std::istringstream s(c); std::string m; s >> m; ssh.CallMethod(m, s); }
    } else {
      std::istringstream s("0.0 -0.105 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 "
                           "0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 "
                           "0.2 -0.21 0.0 0.0 0.21 0.0");
      ssh.ReadStepSequenceAccordingToWalkMode(s);
    }
    std::deque<RelativeFootPosition> rel;
    ssh.CopyRelativeFootPosition(rel, true);
    std::deque<ZMPPosition> zq;
    std::deque<COMState> cq;
    std::deque<FootAbsolutePosition> lq, rq;
    FootAbsolutePosition il, ir;
    memset(&il, 0, sizeof il); memset(&ir, 0, sizeof ir);
    il.x = 0.0094903; il.y = 0.095; ir.x = 0.0094903; ir.y = -0.095;     // EvaluateStartingState on HRP-2's half-sitting pose
    COMState c0;
    double z0[3] = {0, 0, 0};
    zmpd.SetCurrentTime(0.0);
    zmpd.GetZMPDiscretization(zq, cq, rel, lq, rq, 0.0, c0, z0, il, ir);
    if (zq.size() != lq.size() || zq.size() != rq.size() || zq.size() != cq.size()) throw std::runtime_error("deque sizes differ");
    // the same sequence fed on line: InitOnLine with the first steps, OnLineAddFoot for the others, then the end phase
    {
      ZMPDiscretization online(&spm, "", &robot);
      for (const char *c : cmds) { std::istringstream s(c); std::string m; s >> m; spm.CallMethod(m, s); }
      std::deque<RelativeFootPosition> first(rel.begin(), rel.begin() + 3);
      std::deque<ZMPPosition> z2q; std::deque<COMState> c2q; std::deque<FootAbsolutePosition> l2q, r2q;
      online.SetCurrentTime(0.0);
      online.InitOnLine(z2q, c2q, l2q, r2q, il, ir, first, c0, z0);
      for (size_t i = 3; i < rel.size(); i++) online.OnLineAddFoot(rel[i], z2q, c2q, l2q, r2q, false);
      online.EndPhaseOfTheWalking(z2q, c2q, l2q, r2q);
      if (z2q.size() != zq.size()) throw std::runtime_error("on-line and off-line sequences differ in length");
      for (size_t i = 0; i < zq.size(); i++)
        if (z2q[i].px != zq[i].px || z2q[i].py != zq[i].py || l2q[i].x != lq[i].x || l2q[i].z != lq[i].z || r2q[i].y != rq[i].y ||
            z2q[i].time != zq[i].time)
          throw std::runtime_error("on-line and off-line sequences differ");
    }
    const double ex = zq.back().px, ey = zq.back().py;
    const int L = (int)zq.size() - (int)nl + 1;

    // (a) the reference's call pattern for the first 40 control steps
    std::vector<double> x(3, 0.0), y(3, 0.0);
    double sxz = 0.0, syz = 0.0, zx2 = 0.0, zy2 = 0.0;
    std::vector<double> first;
    for (unsigned l = 0; l < 40; l++) {
      pc.OneIterationOfPreview(x, y, sxz, syz, zq, l, zx2, zy2, true);
      first.push_back(x[0]); first.push_back(y[0]); first.push_back(zx2); first.push_back(zy2);
    }
    // (b) the whole run in one launch
    std::vector<double> zx(zq.size()), zy(zq.size()), com((size_t)L * 6), z2((size_t)L * 2);
    for (size_t i = 0; i < zq.size(); i++) { zx[i] = zq[i].px; zy[i] = zq[i].py; }
    double st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    pc.RunBatch(1, L, zx.data(), zy.data(), st, com.data(), z2.data(), true);
    for (unsigned l = 0; l < 40; l++)
      if (first[4 * l] != com[6 * l] || first[4 * l + 1] != com[6 * l + 3] || first[4 * l + 2] != z2[2 * l] || first[4 * l + 3] != z2[2 * l + 1])
        throw std::runtime_error("step-by-step and batched runs differ");
    FILE *f = fopen(out, "w");
    if (!f) throw std::runtime_error("cannot open output file");
    for (int l = 0; l < L; l++)
      fprintf(f, "%.3f %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (l + 1) * T,
              com[6 * l], com[6 * l + 3], z2[2 * l], z2[2 * l + 1], zx[l], zy[l], lq[l].x, lq[l].y, lq[l].z, rq[l].x, rq[l].y,
              rq[l].z, lq[l].theta, rq[l].theta);
    fclose(f);
    printf("TestKajita2003 stage 1: %d control steps, final CoM (%.6f, %.6f), last footprint (%.3f, %.3f)\n", L,
           com[6 * (L - 1)], com[6 * (L - 1) + 3], ex, ey);
  } catch (std::exception &e) {
    std::cerr << "FAILED: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
