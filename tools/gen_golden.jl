# Emits true-reference fixtures in this repository's golden format (tests/golden/c172s0_config1.npz keys),
# for anyone who has Julia 1.12 + Flight.jl. Run from a Flight.jl checkout:
#     julia --project tools/gen_golden.jl out_dir
# Writes raw little-endian Float64 files (one per array, column-major) plus a manifest; tests/golden/
# from_julia.py (not needed until such files exist) would repack them as .npz.
using Flight
out = length(ARGS) > 0 ? ARGS[1] : "."
world = SimpleWorld(; aircraft = Cessna172Sv0()) |> Model
sim = Simulation(world; dt = 0.01, t_end = 10, save_on = false)
init!(sim, C172.TrimParameters())
write(joinpath(out, "x0.f64"), collect(sim.x))
traj = Vector{Vector{Float64}}([collect(sim.x)])
for k in 1:10
    step!(sim, 1.0, true)
    push!(traj, collect(sim.x))
end
write(joinpath(out, "traj.f64"), reduce(hcat, traj))
f_ode!(world)
write(joinpath(out, "xdot0.f64"), collect(world.ẋ))
println("wrote x0 (27), traj (27 x 11, every 100 steps of dt = 0.01), xdot at t_end")
