// test_two_objects.cpp -- the reference keeps its solver set-up per object (one PreviewControl, one
// ZMPVelocityReferencedQP ... per PatternGeneratorInterface, all of them independent).  Here that state lives on the
// device, in a context of the C ABI owned by the facade object.  This program interleaves two objects of each kind that
// differ in their configuration and checks each against the same object run alone: bit-identical.
//   two PreviewControl objects (preview window 1.6 s at zc = 0.814 vs 0.8 s at zc = 0.70), steps alternating
//   two PatternGeneratorInterface objects on different robots (sole size), control loops alternating
#include <cmath>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <vector>

#include "../../include/wg_walkgen.hh"

using namespace PatternGeneratorJRL;
using namespace std;

static void cmd(PatternGeneratorInterface &aPGI, const char *c) { istringstream strm(c); aPGI.ParseCmd(strm); }

static PreviewControl *makePC(SimplePluginManager *spm, double window, double zc) {
  PreviewControl *pc = new PreviewControl(spm, OptimalControllerSolver::MODE_WITHOUT_INITIALPOS, false);
  pc->SetSamplingPeriod(0.005);
  pc->SetPreviewControlTime(window);
  pc->SetHeightOfCoM(zc);
  pc->ComputeOptimalWeights(OptimalControllerSolver::MODE_WITHOUT_INITIALPOS);
  return pc;
}

struct PCRun { vector<double> x, y; double sx, sy; vector<double> trace; PCRun() : x(3, 0.0), y(3, 0.0), sx(0), sy(0) {} };
static void pcStep(PreviewControl *pc, PCRun &r, deque<ZMPPosition> &q, unsigned l) {
  double zx = 0, zy = 0;
  pc->OneIterationOfPreview(r.x, r.y, r.sx, r.sy, q, l, zx, zy, true);
  r.trace.push_back(r.x[0]); r.trace.push_back(r.y[0]); r.trace.push_back(zx); r.trace.push_back(zy);
}

static PatternGeneratorInterface *makePGI(const HumanoidModel *hm) {
  PatternGeneratorInterface *p = patternGeneratorInterfaceFactory(hm);
  const char *init[7] = {":samplingperiod 0.005", ":previewcontroltime 1.6", ":comheight 0.8078", ":SetAlgoForZmpTrajectory Herdt",
                         ":singlesupporttime 0.7", ":doublesupporttime 0.1", ":HerdtOnline"};
  for (int i = 0; i < 7; i++) cmd(*p, init[i]);
  cmd(*p, ":numberstepsbeforestop 2");
  return p;
}
static void pgiRun(PatternGeneratorInterface *p, int from, int to, vector<double> &trace) {
  vectorN q, dq, ddq, zmp(3, 0.0);
  COMState c;
  FootAbsolutePosition L, R;
  for (int it = from; it < to; it++) {
    if (it == 200) p->setVelocityReference(0.2, 0.0, 0.1);
    if (!p->RunOneStepOfTheControlLoop(q, dq, ddq, zmp, c, L, R)) break;
    trace.push_back(c.x[0]); trace.push_back(c.y[0]); trace.push_back(zmp[0]); trace.push_back(L.x); trace.push_back(R.y);
  }
}

int main() {
  try {
    // ---- PreviewControl ----
    SimplePluginManager spm;
    deque<ZMPPosition> q;
    for (int i = 0; i < 900; i++) { ZMPPosition z; memset(&z, 0, sizeof z); z.px = 0.05 * sin(0.01 * i); z.py = (i / 150) % 2 ? 0.09 : -0.09; q.push_back(z); }
    const unsigned L = 400;
    PCRun aloneA, aloneB, mixA, mixB;
    { PreviewControl *a = makePC(&spm, 1.6, 0.814); for (unsigned l = 0; l < L; l++) pcStep(a, aloneA, q, l); delete a; }
    { PreviewControl *b = makePC(&spm, 0.8, 0.70); for (unsigned l = 0; l < L; l++) pcStep(b, aloneB, q, l); delete b; }
    {
      PreviewControl *a = makePC(&spm, 1.6, 0.814), *b = makePC(&spm, 0.8, 0.70);
      for (unsigned l = 0; l < L; l++) { pcStep(a, mixA, q, l); pcStep(b, mixB, q, l); }
      delete a; delete b;
    }
    if (aloneA.trace != mixA.trace || aloneB.trace != mixB.trace) { fprintf(stderr, "FAILED: interleaved PreviewControl objects disturb each other\n"); return 1; }
    if (aloneA.trace == aloneB.trace) { fprintf(stderr, "FAILED: the two PreviewControl configurations do not differ\n"); return 1; }
    // ---- PatternGeneratorInterface on two robots ----
    HumanoidModel r1 = HumanoidModel::sampleRobot(), r2 = HumanoidModel::sampleRobot();
    r2.soleWidth = 0.20; r2.soleHeight = 0.12; r2.startCoM[2] = 0.68;
    const int N = 1200;
    vector<double> alone1, alone2, mix1, mix2;
    { PatternGeneratorInterface *p = makePGI(&r1); pgiRun(p, 0, N, alone1); delete p; }
    { PatternGeneratorInterface *p = makePGI(&r2); pgiRun(p, 0, N, alone2); delete p; }
    {
      PatternGeneratorInterface *p1 = makePGI(&r1), *p2 = makePGI(&r2);      // the second one configured after the first
      for (int it = 0; it < N; it += 40) { pgiRun(p1, it, it + 40, mix1); pgiRun(p2, it, it + 40, mix2); }
      delete p1; delete p2;
    }
    if (alone1 != mix1 || alone2 != mix2) { fprintf(stderr, "FAILED: interleaved PatternGeneratorInterface objects disturb each other\n"); return 1; }
    if (alone1 == alone2) { fprintf(stderr, "FAILED: the two robots do not walk differently\n"); return 1; }
    printf("two objects ok: %zu preview samples, %zu control steps per object, bit-identical to running alone\n", aloneA.trace.size() / 4, alone1.size() / 5);
  } catch (const std::exception &e) {
    fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  return 0;
}
