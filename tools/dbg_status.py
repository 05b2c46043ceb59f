import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "flight.jl_amd"))
import flightbatch as fb
for n in (4096, 65536, 262144, 1 << 20):
    w = fb.BatchedWorld(n)
    fb.f_init(w, fb.TrimParameters(EAS=np.linspace(40, 48, n)))
    for k in (1, 50):
        sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=k)
        fb.step(sim, 0.5); w.sync()
        st = w.status
        nz = np.nonzero(st)[0]
        print(n, "k=", k, "nonzero status:", len(nz), "first idx", nz[:5], "vals", st[nz[:3]])
    w.close()
print("---- with set_state between trim and step ----")
n = 4096
w = fb.BatchedWorld(n)
fb.f_init(w, fb.TrimParameters(EAS=np.linspace(40, 48, n)))
w.set_state(w.x, w.s)
sim = fb.Simulation(w, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 0.5); w.sync()
print("nonzero:", (w.status != 0).sum())
print("---- f_ode then step ----")
w2 = fb.BatchedWorld(n)
fb.f_init(w2, fb.TrimParameters(EAS=np.linspace(40, 48, n)))
fb.f_ode(w2); w2.sync(); print("after f_ode nonzero:", (w2.status != 0).sum())
sim = fb.Simulation(w2, dt=0.01, save_on=False, steps_per_launch=50)
fb.step(sim, 0.5); w2.sync()
print("nonzero:", (w2.status != 0).sum(), w2.status[:4])
