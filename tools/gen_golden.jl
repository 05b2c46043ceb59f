# Emits true-reference fixtures in this repository's golden format (tests/golden/c172s0_config1.npz keys),
# for anyone who has Julia 1.12 + Flight.jl. Run from a Flight.jl checkout:
#     julia --project tools/gen_golden.jl <this repo>/tests/golden/julia
# Writes raw little-endian Float64 files (one per array, column-major). tests/golden/from_julia.py reads them and
# tests/test_julia_fixtures.py then checks the CPU oracle AND the GPU path against them at 1e-6 (it skips while they are absent).
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

# ---- Cessna172Xv2, the scenario of README example 2 (tests/golden/c172x2_modes8.npz, aircraft 0) -------------------------------
# State rows come out in the reference's ComponentVector order, which is the order fb_get_state presents (include/flightbatch.h).
using Flight.FlightApps.C172X.C172XControl: ModeControlLon, ModeControlLat
worldx = SimpleWorld(; aircraft = Cessna172Xv2()) |> Model
worldx.atmosphere.wind.u.N = 1.0; worldx.atmosphere.wind.u.E = 0.5
simx = Simulation(worldx; dt = 0.01, Δt = 0.02, t_end = 20, save_on = false)
init!(simx, C172.TrimParameters())
ctl = worldx.aircraft.avionics.ctl
ctl.u.lon.mode_req = ModeControlLon.EAS_clm; ctl.u.lon.clm_ref = 2.0
ctl.u.lat.mode_req = ModeControlLat.φ_β;     ctl.u.lat.φ_ref = deg2rad(30)
trajx = Vector{Vector{Float64}}([collect(simx.x)])
for k in 1:10
    step!(simx, 2.0, true)
    push!(trajx, collect(simx.x))
end
write(joinpath(out, "x2_traj.f64"), reduce(hcat, trajx))
println("wrote x2_traj (34 x 11, every 200 steps of dt = 0.01, control laws at 0.02 s)")

# ---- Robot2D, mode_v with v_ref = 0.3 (tests/golden/robot2d_modes4.npz, robot 0) ------------------------------------------------
using Flight.FlightApps.Robot2D
robot = Robot2D.Robot() |> Model
simr = Simulation(robot; dt = 0.01, Δt = 0.02, t_end = 10, save_on = false)
init!(simr, Robot2D.InitParameters())
robot.controller.u.mode = Robot2D.mode_v; robot.controller.u.v_ref = 0.3
trajr = Vector{Vector{Float64}}([collect(simr.x)])
for k in 1:10
    step!(simr, 1.0, true)
    push!(trajr, collect(simr.x))
end
write(joinpath(out, "robot2d_traj.f64"), reduce(hcat, trajr))
println("wrote robot2d_traj (4 x 11: ω, v, θ, η every 100 steps)")
