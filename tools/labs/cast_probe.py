import time, torch
n = 1 << 20
x = torch.randint(-1, 2, (n,), dtype=torch.int8, device="cuda")
o = torch.empty(n, dtype=torch.float32, device="cuda")
def T(name, f, K=2000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(K): f()
    torch.cuda.synchronize(); print("%-40s %7.2f us" % (name, (time.perf_counter() - t) / K * 1e6))
T("x.to(float32)", lambda: x.to(torch.float32))
T("o.copy_(x)", lambda: o.copy_(x))
T("torch.mul(x, 1.0, out=o)", lambda: torch.mul(x, 1.0, out=o))
T("x.float()", lambda: x.float())
T("torch.neg(o, out=o)", lambda: torch.neg(o, out=o))
T("x == 0 (bool out)", lambda: x == 0)
y = torch.empty(n, dtype=torch.uint8, device="cuda")
T("x|x", lambda: x | x)

# the same casts interleaved with env steps (what a loop that reads its rewards does)
import sys; sys.path.insert(0, ".")
from gym_soccer_littman94_amd import VectorSoccerEnv
v = VectorSoccerEnv(n, seed=0, io="device"); v.reset()
ta = torch.randint(0, 5, (8, 2, n), dtype=torch.int8, device="cuda")
pairs = [{'player_a': ta[k % 8, 0], 'player_b': ta[k % 8, 1]} for k in range(8)]
kk = [0]
def step():
    kk[0] = (kk[0] + 1) & 7
    return v.step(pairs[kk[0]])
T("env.step", step)
T("env.step + unrelated o.copy_(x)", lambda: (step(), o.copy_(x)))
T("env.step + o.copy_(env._rew)", lambda: (step(), o.copy_(v._rew)))
T("env.step + env._rew.to(float32)", lambda: (step(), v._rew.to(torch.float32)))
T("env.step + r['player_a']", lambda: step()[1]["player_a"])
T("env.step + r['player_a'] + r['player_b']", lambda: (lambda r: (r["player_a"], r["player_b"]))(step()[1]))
T("env.step + info['_final_observation']", lambda: step()[4]["_final_observation"])
T("env.step x2", lambda: (step(), step()))
