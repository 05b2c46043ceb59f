# Times the true reference path (BASELINE.md B2): one Cessna172Sv0, RK4 dt = 0.01, 10 s, no logging.
#     julia --project tools/bench_reference.jl
using Flight
sim = Simulation(Model(SimpleWorld(Cessna172Sv0())); dt = 0.01, save_on = false)
init!(sim, C172.TrimParameters()); step!(sim, 10.0, true)          # warm-up / compile
init!(sim, C172.TrimParameters())
t = @elapsed step!(sim, 10.0, true)
println("reference: $(1000 / t) aircraft-steps/s on 1 core ($(t) s for 1000 steps)")
