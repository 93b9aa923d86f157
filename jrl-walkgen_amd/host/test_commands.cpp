// test_commands.cpp -- every command string the reference's PatternGeneratorInterface registers
// (PatternGeneratorInterfacePrivate.cpp:186-201, dispatch :1055-1133) sent through ParseCmd of this build's interface, plus the
// step-stack generators (StepStackHandler.cpp:929-1040), each checked for its effect -- or for the documented refusal where
// the generator behind it is outside the Herdt-2010 / Kajita stage-1 path.  Prints one line per check; exit code 0 = all passed.
//   test_commands [scratch-directory]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/wg_walkgen.hh"

using namespace PatternGeneratorJRL;

static int g_failed = 0;
static void check(bool ok, const std::string &what) {
  printf("%s  %s\n", ok ? "ok  " : "FAIL", what.c_str());
  if (!ok) g_failed++;
}
static void cmd(PatternGeneratorInterface &pgi, const std::string &c) {
  std::istringstream s(c);
  pgi.ParseCmd(s);
}
static void send(SimplePluginManager &spm, const std::string &c) {      // what ParseCmd does (:1029-1039), on a bare manager
  std::istringstream s(c);
  std::string name;
  s >> name;
  spm.CallMethod(name, s);
}
static HumanoidModel hrp2_like() {
  HumanoidModel robot = HumanoidModel::sampleRobot();
  robot.startLeftFoot[0] = 0.0094903; robot.startLeftFoot[1] = 0.095; robot.startLeftFoot[2] = 0.0;
  robot.startRightFoot[0] = 0.0094903; robot.startRightFoot[1] = -0.095; robot.startRightFoot[2] = 0.0;
  return robot;
}
static PatternGeneratorInterface *fresh(const HumanoidModel &robot) {
  PatternGeneratorInterface *pgi = patternGeneratorInterfaceFactory(const_cast<HumanoidModel *>(&robot));
  const char *common[] = {":comheight 0.8078", ":samplingperiod 0.005", ":previewcontroltime 1.6", ":omega 0.0", ":stepheight 0.07",
                          ":singlesupporttime 0.78", ":doublesupporttime 0.02", ":armparameters 0.5", ":LimitsFeasibility 0.0",
                          ":SetAlgoForZmpTrajectory Kajita"};
  for (const char *c : common) cmd(*pgi, c);
  return pgi;
}
struct Trace { std::vector<double> zx, zy, lx, ly, lth, rx, ry, rth; };
static Trace run(PatternGeneratorInterface &pgi) {
  Trace t;
  vectorN q, dq, ddq, zmp;
  COMState com;
  FootAbsolutePosition lf, rf;
  unsigned long n = 0;
  while (pgi.RunOneStepOfTheControlLoop(q, dq, ddq, zmp, com, lf, rf)) {
    t.zx.push_back(zmp[0]); t.zy.push_back(zmp[1]);
    t.lx.push_back(lf.x); t.ly.push_back(lf.y); t.lth.push_back(lf.theta);
    t.rx.push_back(rf.x); t.ry.push_back(rf.y); t.rth.push_back(rf.theta);
    if (++n > 200000) throw std::runtime_error("the control loop does not end");
  }
  return t;
}
static bool same(const Trace &a, const Trace &b) {
  return a.zx == b.zx && a.zy == b.zy && a.lx == b.lx && a.ly == b.ly && a.rx == b.rx && a.ry == b.ry && a.lth == b.lth && a.rth == b.rth;
}

int main(int argc, char **argv) {
  const std::string scratch = argc > 1 ? argv[1] : "/tmp";
  try {
    const HumanoidModel robot = hrp2_like();
    const std::string walk = "0.2 0.21 0.0 0.2 -0.21 0.0 0.2 0.21 0.0 0.2 -0.21 0.0 0.0 0.21 0.0";

    // ---- reference run: an explicit first half step ----
    Trace base;
    {
      PatternGeneratorInterface *pgi = fresh(robot);
      cmd(*pgi, ":stepseq 0.0 -0.105 0.0 " + walk);
      base = run(*pgi);
      check(base.zx.size() > 1000, ":stepseq / :SetAlgoForZmpTrajectory / :samplingperiod / :LimitsFeasibility: a walk of " +
                                       std::to_string(base.zx.size()) + " control steps");
      delete pgi;
    }
    // ---- :ZMPShiftParameters reaches ZMPDiscretization (only step types 3, 4, 5 read it: a plain walk must not move) and
    //      :TimeDistributionParameters is accepted ----
    {
      PatternGeneratorInterface *pgi = fresh(robot);
      cmd(*pgi, ":ZMPShiftParameters 0.015 0.015 0.015 0.015");
      cmd(*pgi, ":TimeDistributionParameters 2.0 3.5 1.0 3.0");
      cmd(*pgi, ":stepseq 0.0 -0.105 0.0 " + walk);
      check(same(run(*pgi), base), ":ZMPShiftParameters / :TimeDistributionParameters accepted; a walk of plain steps is unchanged");
      delete pgi;
    }
    {
      // the shift itself, at the generator it belongs to: obstacle step types read it (ZMPDiscretization.cpp:693-718)
      SimplePluginManager spm;
      ZMPDiscretization a(&spm, "", &robot), b(&spm, "", &robot);
      std::vector<double> shift(4, 0.015);
      b.SetZMPShift(shift);
      for (const char *c : {":samplingperiod 0.005", ":previewcontroltime 1.6", ":singlesupporttime 0.78", ":doublesupporttime 0.02",
                            ":stepheight 0.07", ":omega 0.0", ":comheight 0.8078"})
        send(spm, c);
      auto produce = [&](ZMPDiscretization &z) {
        std::deque<ZMPPosition> zp; std::deque<COMState> cs; std::deque<FootAbsolutePosition> l, r;
        std::deque<RelativeFootPosition> steps;
        const double seq[][3] = {{0.0, -0.105, 0.0}, {0.2, 0.21, 0.0}, {0.2, -0.21, 0.0}, {0.2, 0.21, 0.0}, {0.0, -0.21, 0.0}};
        for (int i = 0; i < 5; i++) {
          RelativeFootPosition f; memset(&f, 0, sizeof f);
          f.sx = seq[i][0]; f.sy = seq[i][1]; f.theta = seq[i][2]; f.SStime = 0.78; f.DStime = 0.02; f.stepType = (i == 2) ? 3 : 1;
          steps.push_back(f);
        }
        COMState c0; c0.z[0] = 0.8078;
        double z0[3] = {0, 0, 0};
        FootAbsolutePosition il, ir; memset(&il, 0, sizeof il); memset(&ir, 0, sizeof ir);
        il.x = robot.startLeftFoot[0]; il.y = robot.startLeftFoot[1]; ir.x = robot.startRightFoot[0]; ir.y = robot.startRightFoot[1];
        z.GetZMPDiscretization(zp, cs, steps, l, r, 0.0, c0, z0, il, ir);
        std::vector<double> out;
        for (auto &p : zp) out.push_back(p.px);
        return out;
      };
      const std::vector<double> za = produce(a), zb = produce(b);
      double dmax = 0.0;
      for (size_t i = 0; i < za.size() && i < zb.size(); i++) dmax = std::fmax(dmax, std::fabs(za[i] - zb[i]));
      check(za.size() == zb.size() && dmax > 1e-3 && dmax < 0.1, "SetZMPShift moves the ZMP of an obstacle-type step (max |dx| = " + std::to_string(dmax) + ")");
    }
    // ---- :SetAutoFirstStep true: the first half step is derived from the start state (AutomaticallyAddFirstStep) ----
    {
      PatternGeneratorInterface *pgi = fresh(robot);
      cmd(*pgi, ":SetAutoFirstStep true");
      cmd(*pgi, ":stepseq " + walk);
      const Trace t = run(*pgi);
      // CoM (startCoM) -> right foot: the step the walk above would otherwise have had to spell out
      PatternGeneratorInterface *ref = fresh(robot);
      char first[128];
      snprintf(first, sizeof first, "%.17g %.17g 0.0 ", robot.startRightFoot[0] - robot.startCoM[0], robot.startRightFoot[1] - robot.startCoM[1]);
      cmd(*ref, std::string(":stepseq ") + first + walk);
      check(same(t, run(*ref)), ":SetAutoFirstStep true = the same walk with the step (CoM -> first support foot) spelled out");
      PatternGeneratorInterface *off = fresh(robot);
      cmd(*off, ":SetAutoFirstStep true"); cmd(*off, ":SetAutoFirstStep false");
      cmd(*off, ":stepseq 0.0 -0.105 0.0 " + walk);
      check(same(run(*off), base), ":SetAutoFirstStep false restores the default");
      delete pgi; delete ref; delete off;
    }
    // ---- on-line step sequencing belongs to the step-stack generators outside this path ----
    {
      PatternGeneratorInterface *pgi = fresh(robot);
      bool refused = false;
      try { cmd(*pgi, ":StartOnLineStepSequencing 0.0 -0.105 0.0 " + walk); } catch (const NotOnThisPath &) { refused = true; }
      check(refused, ":StartOnLineStepSequencing reads its steps, then refuses loudly (NotOnThisPath)");
      cmd(*pgi, ":StopOnLineStepSequencing");
      cmd(*pgi, ":ChangeNextStep 1.0 0.2 0.19 0.0");
      cmd(*pgi, ":readfilefromkw /nonexistent/path.kw 1");
      cmd(*pgi, ":finish");                               // the steps read above are still on the stack
      check(same(run(*pgi), base), ":StopOnLineStepSequencing / :ChangeNextStep / :readfilefromkw change nothing; :finish runs the stacked steps");
      delete pgi;
    }
    // ---- :arccentered: every arc step turns by the same angle and lands on its circle around the centre ----
    {
      PatternGeneratorInterface *pgi = fresh(robot);
      cmd(*pgi, ":supportfoot -1");
      cmd(*pgi, ":arccentered 0.75 30.0 -1");
      cmd(*pgi, ":lastsupport");
      cmd(*pgi, ":finish");
      const Trace t = run(*pgi);
      const double th_end = 0.5 * (t.lth.back() + t.rth.back());
      check(t.zx.size() > 1000 && std::fabs(th_end - 30.0) < 1e-6, ":arccentered 0.75 30 -1: " + std::to_string(t.zx.size()) +
                                                                       " control steps, final feet heading " + std::to_string(th_end) + " deg");
      // Where a foot rests (consecutive equal samples) is a footprint.  Each arc step turns both feet by the same angle about the
      // same centre, so the rigid motion from one footprint of a foot to its next one, seen from the first, is the same every
      // time (all but the last, shorter step): 0.10 m of arc at radius 0.75 m = 7.639 degrees.
      double worst = 0.0, dth_seen = 0.0;
      int pairs = 0;
      for (int foot = 0; foot < 2; foot++) {
        const std::vector<double> &X = foot ? t.rx : t.lx, &Y = foot ? t.ry : t.ly, &TH = foot ? t.rth : t.lth;
        std::vector<std::vector<double> > prints;
        for (size_t i = 0; i + 1 < X.size(); i++) {
          if (X[i] != X[i + 1] || Y[i] != Y[i + 1] || TH[i] != TH[i + 1]) continue;         // in flight
          if (prints.empty() || prints.back()[0] != X[i] || prints.back()[1] != Y[i] || prints.back()[2] != TH[i])
            prints.push_back({X[i], Y[i], TH[i]});
        }
        std::vector<std::vector<double> > rel;
        for (size_t k = 0; k + 1 < prints.size(); k++) {
          const double th = prints[k][2] * 3.14159265358979323846 / 180.0, c = std::cos(th), s = std::sin(th);
          const double dx = prints[k + 1][0] - prints[k][0], dy = prints[k + 1][1] - prints[k][1];
          rel.push_back({c * dx + s * dy, -s * dx + c * dy, prints[k + 1][2] - prints[k][2]});
        }
        // whole arc steps: those that turn by the full step angle
        for (size_t k = 1; k + 1 < rel.size(); k++) {               // rel[0] starts from the rest posture, not from a footprint of the arc
          if (std::fabs(rel[k][2] - 7.6394372684109761) > 1e-6 || std::fabs(rel[k + 1][2] - 7.6394372684109761) > 1e-6) continue;
          for (int e = 0; e < 3; e++) worst = std::fmax(worst, std::fabs(rel[k][e] - rel[k + 1][e]));
          dth_seen = rel[k][2];
          pairs++;
        }
      }
      check(pairs >= 2 && worst < 1e-9, ":arccentered: successive footprints of a foot differ by one and the same rigid motion (" +
                                            std::to_string(pairs) + " pairs, spread " + std::to_string(worst) + ", turn " + std::to_string(dth_seen) + " deg)");
      delete pgi;
    }
    // ---- Herdt mode: the failed-QP dump (ZMPVelocityReferencedQP.cpp:399-402) and the dump on request ----
    {
      SimplePluginManager spm;
      HumanoidModel r2 = HumanoidModel::sampleRobot();
      ZMPVelocityReferencedQP qp(&spm, "", &r2);
      send(spm, ":samplingperiod 0.005");
      std::deque<ZMPPosition> zp; std::deque<COMState> cs; std::deque<FootAbsolutePosition> l, r; std::deque<RelativeFootPosition> rel;
      COMState c0;
      c0.x[0] = r2.startCoM[0]; c0.y[0] = r2.startCoM[1]; c0.z[0] = r2.startCoM[2];
      FootAbsolutePosition il, ir; memset(&il, 0, sizeof il); memset(&ir, 0, sizeof ir);
      il.x = r2.startLeftFoot[0]; il.y = r2.startLeftFoot[1]; ir.x = r2.startRightFoot[0]; ir.y = r2.startRightFoot[1];
      double z0[3] = {0, 0, 0};
      qp.SetCurrentTime(0.0);
      qp.InitOnLine(zp, cs, l, r, il, ir, rel, c0, z0);
      qp.Reference(0.2, 0.0, 0.0);
      double time = 0.0;
      for (int k = 0; k < 400; k++) { time += 0.005; qp.OnLine(time, zp, cs, l, r); }
      check(qp.Solution().Fail == 0, "Herdt mode: 2 s of walking, every QP solved");
      const std::string file = scratch + "/wg_problem_on_request.dat";
      qp.dumpProblem(file.c_str());
      std::ifstream in(file.c_str());
      std::string tok;
      int nq = 0, mq = 0, rows_du = 0, cols_du = 0;
      bool sym = true, have_params = false;
      if (in >> tok && sscanf(tok.c_str(), "Q[%d,%d]", &nq, &mq) == 2 && nq == mq && nq >= 32 && nq <= 40) {
        std::vector<double> Q((size_t)nq * nq);
        for (int i = 0; i < nq; i++) for (int j = 0; j < nq; j++) in >> Q[(size_t)i * nq + j];
        for (int i = 0; i < nq; i++) for (int j = 0; j < nq; j++) sym = sym && std::fabs(Q[(size_t)i * nq + j] - Q[(size_t)j * nq + i]) < 1e-15;
        int one = 0, dn = 0;
        in >> tok; sscanf(tok.c_str(), "D[%d,%d]", &dn, &one);
        double v; for (int i = 0; i < dn; i++) in >> v;
        in >> tok; sscanf(tok.c_str(), "DU[%d,%d]", &rows_du, &cols_du);
        std::string rest((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        have_params = rest.find("XU[") != std::string::npos && rest.find("mmax: " + std::to_string(rows_du)) != std::string::npos &&
                      rest.find("Eps: 1e-08") != std::string::npos;
      }
      check(nq >= 32 && sym && cols_du == nq && rows_du >= 66 && have_params,
            "dumpProblem writes QPProblem::dump_problem's layout: Q[" + std::to_string(nq) + "," + std::to_string(nq) + "] symmetric, DU[" +
                std::to_string(rows_du) + "," + std::to_string(cols_du) + "], DS, XL, XU, solver parameters");
      // The dump OnLine writes for a tick (WG_DUMP_EVERY_QP; WG_DUMP_FAILED_QP writes it only when Solution().Fail > 0, as the
      // reference does) is the problem of THAT tick: the same bytes as a dump taken on request just before the tick fired.
      const std::string before_file = scratch + "/wg_problem_before_tick.dat";
      double t_tick = 0.0;
      setenv("WG_DUMP_EVERY_QP", scratch.c_str(), 1);
      for (int k = 0; k < 40 && t_tick == 0.0; k++) {
        const int ticks_before = qp.State().tick_count;
        qp.dumpProblem(before_file.c_str());
        time += 0.005;
        qp.OnLine(time, zp, cs, l, r);
        if (qp.State().tick_count != ticks_before) t_tick = time;
      }
      unsetenv("WG_DUMP_EVERY_QP");
      char name[1024];
      snprintf(name, sizeof name, "%s/Problem_%f.dat", scratch.c_str(), t_tick);
      auto slurp = [](const std::string &f) { std::ifstream i(f.c_str()); return std::string((std::istreambuf_iterator<char>(i)), std::istreambuf_iterator<char>()); };
      const std::string a = slurp(name), b = slurp(before_file);
      check(t_tick > 0.0 && a.size() > 10000 && a == b, std::string("the tick's own problem is what OnLine dumps: ") + name + " (" + std::to_string(a.size()) + " bytes)");
    }
  } catch (std::exception &e) {
    std::cerr << "FAILED: " << e.what() << std::endl;
    return 1;
  }
  printf("%s\n", g_failed ? "SOME CHECKS FAILED" : "all checks passed");
  return g_failed ? 2 : 0;
}
