// test_kajita_preview.cpp -- BASELINE config[0] plumbing: the step sequence of the reference's
// TestKajita2003 StraightWalking (tests/TestKajita2003.cpp:96-124) through stage 1 only -- a ZMP reference queue under
// the stance feet, then PreviewControl (include/wg_walkgen.hh) -- on the GPU.  The first steps go through
// OneIterationOfPreview one call at a time (the reference's call pattern, ZMPPreviewControlWithMultiBodyZMP.cpp), the
// whole run through RunBatch; both must agree bit for bit.  Writes "t com_x com_y zmp_x zmp_y zmpref_x zmpref_y".
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

    // ":stepseq 0.0 -0.105 0.0  0.2 0.21 0.0  0.2 -0.21 0.0 ... 0.0 0.21 0.0" (16 triples), single support 0.78 s,
    // double support 0.02 s (CommonInitialization), ZMP under the stance foot, linear hand-over in double support
    const int nsteps = 16;
    double sx[nsteps], sy[nsteps];
    sx[0] = 0.0; sy[0] = -0.105;
    for (int k = 1; k < nsteps - 1; k++) { sx[k] = 0.2; sy[k] = (k % 2) ? 0.21 : -0.21; }
    sx[nsteps - 1] = 0.0; sy[nsteps - 1] = 0.21;
    std::deque<ZMPPosition> zq;
    auto push = [&](double x, double y) { ZMPPosition z; memset(&z, 0, sizeof z); z.px = x; z.py = y; z.time = zq.size() * T; zq.push_back(z); };
    const int n_ss = (int)(0.78 / T + 0.5), n_ds = (int)(0.02 / T + 0.5), n_rest = (int)(1.6 / T + 0.5);
    for (int i = 0; i < n_rest; i++) push(0.0, 0.0);
    double fx = 0.0, fy = 0.0, px = 0.0, py = 0.0;
    for (int k = 0; k < nsteps; k++) {
      fx += sx[k]; fy += sy[k];
      for (int i = 0; i < n_ds; i++) { const double a = (i + 1.0) / n_ds; push(px + a * (fx - px), py + a * (fy - py)); }
      for (int i = 0; i < n_ss; i++) push(fx, fy);
      px = fx; py = fy;
    }
    const double ex = fx, ey = fy - 0.105;
    for (int i = 0; i < n_ds; i++) { const double a = (i + 1.0) / n_ds; push(px + a * (ex - px), py + a * (ey - py)); }
    for (int i = 0; i < n_rest + (int)nl; i++) push(ex, ey);
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
      fprintf(f, "%.3f %.17g %.17g %.17g %.17g %.17g %.17g\n", (l + 1) * T, com[6 * l], com[6 * l + 3], z2[2 * l], z2[2 * l + 1], zx[l], zy[l]);
    fclose(f);
    printf("TestKajita2003StraightWalking stage 1: %d control steps, final CoM (%.6f, %.6f), last footprint (%.3f, %.3f)\n", L,
           com[6 * (L - 1)], com[6 * (L - 1) + 3], ex, ey);
  } catch (std::exception &e) {
    std::cerr << "FAILED: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
