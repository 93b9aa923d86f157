"""Event simulation of the multi-tick kernel's hand-over policy on measured service times (tools/xrun_stats.py PSAVE=...): one XCD =
G gaits, W waves; a tick of gait g at index t takes c0 + c1 * iterations[t, g] microseconds.  Policies:
  fifo      today's ring: a finished gait goes to the back
  keep<k>   the wave keeps its gait while the gait is more than k ticks behind the XCD's mean progress (no hand-over at all then)
  oldest    ideal: always the waiting gait with the fewest ticks done
Prints the makespan against the work bound (sum of service / W)."""
import heapq, sys, numpy as np
f = np.load(sys.argv[1]); its = f["its"].astype(float); xcd = f["xcd"]
T, B = its.shape
c0, c1 = -30.0, 17.0
W = 256

def simulate(svc, policy, k=1):
    T, G = svc.shape
    done = np.zeros(G, int)
    from collections import deque
    ring = deque(range(G))                                   # fresh gaits first, in order
    heap = []                                                # (finish time, wave, gait)
    now = 0.0; pushes = 0; idle_waves = 0
    waiting_oldest = None
    def take():
        if policy == "oldest":
            if not ring: return None
            g = min(ring, key=lambda q: done[q]); ring.remove(g); return g
        return ring.popleft() if ring else None
    for w in range(W):
        g = take()
        if g is None: break
        heapq.heappush(heap, (svc[0, g], w, g))
    exits = []
    while heap:
        now, w, g = heapq.heappop(heap)
        done[g] += 1
        nxt = None
        if done[g] < T:
            if policy.startswith("keep") and done[g] * G + k * G <= pushes:
                nxt = g                                      # behind the mean progress: go on with it
                pushes += 1
            else:
                ring.append(g); pushes += 1
        if nxt is None: nxt = take()
        if nxt is None: exits.append(now); continue
        heapq.heappush(heap, (now + svc[done[nxt], nxt], w, nxt))
    return now, np.array(exits)

for x in range(2):
    gaits = np.nonzero(xcd[0] == x)[0]
    svc = c0 + c1 * its[:, gaits]
    bound = svc.sum() / W
    print("XCD %d: %d gaits, work bound %.2f ms" % (x, len(gaits), bound / 1e3))
    for pol, k in (("fifo", 0), ("keep", 0), ("keep", 1), ("keep", 2), ("keep", 4), ("oldest", 0)):
        if pol == "oldest" and T > 60: continue
        mk, ex = simulate(svc, pol, k)
        print("   %-7s k=%d: makespan %.2f ms (+%.2f %%), mean idle tail %.2f ms" % (pol, k, mk / 1e3, 100 * (mk / bound - 1), (mk - ex).mean() / 1e3))
