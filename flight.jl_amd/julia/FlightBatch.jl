"""
    FlightBatch

Julia-side binding of `libflightbatch` (include/flightbatch.h): the drop-in for the hot path of
`Model(SimpleWorld(Cessna172Sv0()))` — `f_ode!` / `f_step!` / `f_periodic!` / `init!` / `step!` — executed for N
independent aircraft on one MI355X. It follows the only in-tree FFI precedent of Flight.jl, the plain
`ccall((:sym, lib), Ret, (ArgTypes...), args...)` pattern of `lib/FlightCore/src/joysticks.jl:45-53`.

NOT EXECUTED in this repository's CI (no Julia toolchain in the build environment); the Python package
`flightbatch` is the executed mirror of exactly these calls. Array convention of the C ABI: column-major
`[N x Nfield]`, aircraft index fastest — i.e. a Julia `Matrix{Float64}(undef, N, Nfield)` passes as is.

Usage (what a Flight.jl maintainer would write):

    using Flight, FlightBatch
    world  = FlightBatch.BatchedWorld(1_048_576)              # ≙ Model(SimpleWorld(Cessna172Sv0())) x N
    sim    = FlightBatch.BatchedSimulation(world; dt = 0.01)   # ≙ Simulation(world; dt = 0.01)
    FlightBatch.init!(sim, C172.TrimParameters())               # ≙ init!(sim, C172.TrimParameters())
    FlightBatch.step!(sim, 10.0, true)                          # ≙ step!(sim, 10.0, true)
    x = FlightBatch.state(world)                                # N x 27, same component order as world.x
"""
module FlightBatch

using Flight.FlightCore.Modeling: ModelDefinition
import Flight.FlightCore.Modeling: f_init!, f_ode!, f_step!, f_periodic!
using Flight: FlightPhysics
using Flight.FlightApps: C172
using Flight.FlightPhysics: Propellers, Piston, Control
using Flight.FlightPhysics.Kinematics: WA, ECEF, NED                   # kinematic descriptors (exported at FP/kinematics.jl:11)
using Flight.FlightApps.C172.C172S.C172Sv0: Cessna172Sv0               # FA/c172/c172s/c172s0.jl:9,14-18
using Flight.FlightApps.C172.C172X.C172Xv2: Cessna172Xv2               # FA/c172/c172x/c172x2.jl:12,54-59

const lib = get(ENV, "FLIGHTBATCH_LIB", "libflightbatch")

# layout constants of include/flightbatch.h
const NX, NS, NU, NY, NTP, NTS = 27, 2, 16, 174, 18, 7
const TABLE_EGM96, TABLE_PROPELLER, TABLE_PISTON, TABLE_AERO = Cint(0), Cint(1), Cint(2), Cint(3)

struct Params   # fb_params
    dt::Cdouble; periodic_n::Cint; surface::Cint; T_sl::Cdouble; p_sl::Cdouble
    wind_ned::NTuple{3,Cdouble}; h_terrain::Cdouble
end

check(rc::Integer) = rc == 0 || error(unsafe_string(ccall((:fb_last_error, lib), Cstring, ())))

const MODEL_C172S0, MODEL_C172X2, MODEL_ROBOT2D = Cint(0), Cint(1), Cint(2)
const KIN = Dict(WA => Cint(0), ECEF => Cint(1), NED => Cint(2))        # FB_KIN_*: the vehicle's kinematic descriptor
const TABLE_CTL_GAINS = Cint(5)
const NCU, NCS = 28, 66                                                 # FB_NCU, FB_NCS (Cessna172Xv2 control laws)

"N instances of SimpleWorld(aircraft) resident on one GPU (the batched counterpart of a root Model).
`aircraft` is `Cessna172Sv0(kin)` or `Cessna172Xv2(kin)` (kin = WA(), ECEF() or NED()); its type picks the model id."
mutable struct BatchedWorld <: ModelDefinition
    handle::Ptr{Cvoid}
    n::Int
    nx::Int
    function BatchedWorld(n::Integer, aircraft = Cessna172Sv0(); device::Integer = 0)
        model = aircraft isa Cessna172Xv2 ? MODEL_C172X2 : MODEL_C172S0
        kin = KIN[typeof(aircraft.vehicle.kinematics)]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:fb_create, lib), Cint, (Cint, Cint, Cint, Int64, Cint, Ptr{Ptr{Cvoid}}), model, kin, 0, n, device, h))
        nx = Ref{Cint}(0)
        check(ccall((:fb_dims, lib), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}), h[], nx, C_NULL, C_NULL, C_NULL))
        w = new(h[], n, nx[])      # 27 (WA) / 26 (ECEF) / 24 (NED), Xv2 seven more: length(Model(aircraft).x)
        finalizer(w -> ccall((:fb_destroy, lib), Cint, (Ptr{Cvoid},), w.handle), w)
        upload_tables!(w)
        # Xv2: the ten gain lookups of c172x/control/data packed as include/flightbatch.h documents (flightbatch/ctl_gains.py)
        aircraft isa Cessna172Xv2 && set_table!(w, TABLE_CTL_GAINS, pack_ctl_gains(aircraft.avionics.ctl))
        return w
    end
end

# ---- FB_TABLE_CTL_GAINS: the ten gain-scheduling lookups of ControlLawsLon / ControlLawsLat (c172x_ctl.jl:203-213, 814-819), each a
# PIDData / LQRData of `extrapolate(scale(interpolate(points, BSpline(Linear())), EAS_range, h_range), Flat())` objects
# (FP/control.jl:950-972). Blob layout (include/flightbatch.h; executed twin: flightbatch/ctl_gains.py `_pack`): per lookup a 6-double
# header [nE, nH, EAS_lo, EAS_hi, h_lo, h_hi], then for every grid point — h slowest, EAS next — one record:
#   LQR: vec(K_fbk) (column-major NU x NX), vec(K_fwd), vec(K_int), x_trim, u_trim, z_trim      PID: k_p, k_i, k_d, τ_f
scaled_coefs(e) = e.itp.itp.coefs            # Extrapolation -> ScaledInterpolation -> BSplineInterpolation (Linear(): coefs == the data array)
scaled_ranges(e) = e.itp.ranges
function pack_lookup(fields::Vector)
    rE, rH = scaled_ranges(fields[1])
    nE, nH = length(rE), length(rH)
    blob = Float64[nE, nH, first(rE), last(rE), first(rH), last(rH)]
    for j in 1:nH, i in 1:nE, f in fields
        @assert scaled_ranges(f) == (rE, rH)
        append!(blob, vec(collect(Float64, scaled_coefs(f)[i, j])))      # SMatrix / SVector: column-major; scalar: one value
    end
    return blob
end
pack_lookup(l::Control.LQRData) = pack_lookup(Any[l.K_fbk, l.K_fwd, l.K_int, l.x_trim, l.u_trim, l.z_trim])
pack_lookup(l::Control.PIDData) = pack_lookup(Any[l.k_p, l.k_i, l.k_d, l.τ_f])
"blob order of include/flightbatch.h: te2te, tv2te, vh2te, q2e, c2θ, v2t (lon), ar2ar, φβ2ar, p2φ, χ2φ (lat)"
function pack_ctl_gains(ctl)::Vector{Float64}
    (; lon, lat) = ctl
    vcat(map(pack_lookup, (lon.te2te_lookup, lon.tv2te_lookup, lon.vh2te_lookup, lon.q2e_lookup, lon.c2θ_lookup, lon.v2t_lookup,
                           lat.ar2ar_lookup, lat.φβ2ar_lookup, lat.p2φ_lookup, lat.χ2φ_lookup))...)
end

# avionics.ctl.u / avionics.gdc.u and the control laws' record, as N x NCU / N x NCS matrices (columns = FB_CU_* / FB_CS_*)
ctl_inputs(w::BatchedWorld) = (cu = Matrix{Float64}(undef, w.n, NCU);
    check(ccall((:fb_get_ctl_inputs, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, cu)); cu)
ctl_inputs!(w::BatchedWorld, cu::Matrix{Float64}) = check(ccall((:fb_set_ctl_inputs, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, cu))
ctl_record(w::BatchedWorld) = (cs = Matrix{Float64}(undef, w.n, NCS);
    check(ccall((:fb_get_ctl_state, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, cs)); cs)

function set_table!(w::BatchedWorld, kind::Cint, data::Array)
    dims = Int64[size(data)...]
    check(ccall((:fb_set_table, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Int64}, Cint), w.handle, kind, data, dims, length(dims)))
end

"Hand the library the very tables Flight.jl builds at construction time (SURVEY.md Appendix B)."
function upload_tables!(w::BatchedWorld)
    # EGM96: the Float32 721 x 1441 grid behind Geodesy.egm96_interp (geodesy.jl:186-198)
    egm = Matrix{Float32}(undef, 721, 1441)
    # the file Geodesy itself reads: joinpath(dirname(@__FILE__), "data", "ww15mgh_le.bin") in FP/geodesy.jl:166, i.e. next to the
    # package's entry file (pathof works on the package module FlightPhysics, not on its submodule Geodesy)
    read!(joinpath(dirname(pathof(FlightPhysics)), "data", "ww15mgh_le.bin"), egm)
    set_table!(w, TABLE_EGM96, egm)
    # propeller: Lookup(2, Blade()).data, six 21 x 21 x 1 arrays (propellers.jl:235-250) -> [21, 21, 6]
    lookup = Propellers.Lookup(2, Propellers.Blade())
    d = lookup.data
    prop = cat((dropdims(getfield(d, f); dims = 3) for f in (:C_Fx, :C_Mx, :C_Fz_α, :C_Mz_α, :C_P, :η_p))...; dims = 3)
    set_table!(w, TABLE_PROPELLER, prop)
    # piston and aero blobs are packed in the csrc/tables.h layout by the helpers below
    set_table!(w, TABLE_PISTON, pack_piston(Piston.PistonEngineLookup(300 / 2700, 3100 / 2700)))
    set_table!(w, TABLE_AERO, pack_aero(C172.aero_lookup))
end

# The two packers read the interpolation objects' knots / coefs and lay them out as csrc/tables.h documents (AT_* / PT_* offsets,
# 0-based, in doubles; 2-D tables column-major like the Julia arrays); flightbatch/tables.py (`piston_blob`, `aero_blob`) is the
# executed twin and tests/test_host_and_abi.py checks it against the oracle's builders.
#   linear_interpolation(knots, data, extrapolation_bc = ...) == extrapolate(interpolate(knots, data, Gridded(Linear())), ...):
#       .itp.knots :: Tuple of knot vectors, .itp.coefs :: the data array
#   extrapolate(scale(interpolate(A, BSpline(Linear())), ranges...), Line()): .itp.ranges, .itp.itp.coefs (see scaled_coefs above)
gridded_knots(e) = e.itp.knots
gridded_coefs(e) = e.itp.coefs
put_at!(b::Vector{Float64}, off0::Integer, a) = (v = vec(collect(Float64, a)); b[off0 + 1 : off0 + length(v)] .= v; b)

# csrc/tables.h PT_* offsets
const PT_DELTA_WOT_V, PT_MU_WOT_V, PT_PISTD_N_K, PT_PISTD_MU_K, PT_PISTD_V, PT_PIWOT_N_K, PT_PIWOT_D_K, PT_PIWOT_V = 0, 18, 36, 49, 52, 91, 96, 99
const PT_F_K, PT_PI_RATIO_V, PT_SFC_RATIO_V, PT_SFC_N_K, PT_SFC_PI_K, PT_SFC_POW_V, PT_SIZE = 114, 125, 136, 147, 152, 160, 200
"PistonEngineLookup (FP/piston.jl:60-195) -> the FB_TABLE_PISTON blob"
function pack_piston(l)::Vector{Float64}
    b = zeros(PT_SIZE)
    # δ_wot(n, μ), μ_wot(n, δ): scaled B-splines on fixed ranges the kernels know (piston.jl:78-79, 92-93); only the 2 x 9 samples travel
    @assert scaled_ranges(l.δ_wot) == (range(0.667, 1, length = 2), range(0.401, 0.936, length = 9))
    @assert scaled_ranges(l.μ_wot) == (range(0.667, 1, length = 2), range(0.441, 1, length = 9))
    put_at!(b, PT_DELTA_WOT_V, scaled_coefs(l.δ_wot)); put_at!(b, PT_MU_WOT_V, scaled_coefs(l.μ_wot))
    n13, μ3 = gridded_knots(l.π_std)                       # π_std(n, μ): 13 x 3, Flat (piston.jl:110-133)
    put_at!(b, PT_PISTD_N_K, n13); put_at!(b, PT_PISTD_MU_K, μ3); put_at!(b, PT_PISTD_V, gridded_coefs(l.π_std))
    n5, δ3 = gridded_knots(l.π_wot)                        # π_wot(n, δ): 5 x 3 (piston.jl:140-149)
    put_at!(b, PT_PIWOT_N_K, n5); put_at!(b, PT_PIWOT_D_K, δ3); put_at!(b, PT_PIWOT_V, gridded_coefs(l.π_wot))
    (f11,) = gridded_knots(l.π_ratio)                      # π_ratio(f), sfc_ratio(f): 11 knots shared (piston.jl:157-172)
    @assert gridded_knots(l.sfc_ratio)[1] == f11
    put_at!(b, PT_F_K, f11); put_at!(b, PT_PI_RATIO_V, gridded_coefs(l.π_ratio)); put_at!(b, PT_SFC_RATIO_V, gridded_coefs(l.sfc_ratio))
    nsfc, πsfc = gridded_knots(l.sfc_pow)                  # sfc_pow(n, π): 5 x 8, Line (piston.jl:179-189)
    put_at!(b, PT_SFC_N_K, nsfc); put_at!(b, PT_SFC_PI_K, πsfc); put_at!(b, PT_SFC_POW_V, gridded_coefs(l.sfc_pow))
    return b
end

# csrc/tables.h AT_* offsets
const AT_GE_K, AT_CD_GE_V, AT_CL_GE_V, AT_DF4_K, AT_CD_DF_V, AT_CL_DF_V, AT_CM_DF_V, AT_UNIT3_K, AT_CD_DE_V, AT_CD_BETA_V = 0, 13, 26, 39, 43, 47, 51, 55, 58, 61
const AT_CD_ALPHA_K, AT_CD_ALPHA_DF_V, AT_CY_BETA_K, AT_DF2_K, AT_CY_BETA_DF_V, AT_ALPHA2_K, AT_CY_P_V, AT_CY_R_V, AT_CL_R_V = 64, 90, 194, 197, 199, 205, 207, 211, 215
const AT_CL_ALPHA_K, AT_CL_ALPHA_V, AT_SCALARS, AT_SIZE = 219, 236, 270, 291
"C172.aero_lookup (FA/c172/c172.jl:51-205: NamedTuple of scalars and Gridded(Linear()) / Flat() interpolations) -> the FB_TABLE_AERO blob"
function pack_aero(l)::Vector{Float64}
    (; C_D, C_Y, C_L, C_l, C_m, C_n) = l
    b = zeros(AT_SIZE)
    k1(e) = gridded_knots(e)[1]
    @assert k1(C_D.ge) == k1(C_L.ge)                        # ground-effect tables share their 13 knots (c172.jl:67, 117)
    put_at!(b, AT_GE_K, k1(C_D.ge)); put_at!(b, AT_CD_GE_V, gridded_coefs(C_D.ge)); put_at!(b, AT_CL_GE_V, gridded_coefs(C_L.ge))
    @assert k1(C_D.δf) == k1(C_L.δf) == k1(C_m.δf) == gridded_knots(C_D.α_δf)[2]
    put_at!(b, AT_DF4_K, k1(C_D.δf)); put_at!(b, AT_CD_DF_V, gridded_coefs(C_D.δf)); put_at!(b, AT_CL_DF_V, gridded_coefs(C_L.δf)); put_at!(b, AT_CM_DF_V, gridded_coefs(C_m.δf))
    @assert k1(C_D.δe) == k1(C_D.β)
    put_at!(b, AT_UNIT3_K, k1(C_D.δe)); put_at!(b, AT_CD_DE_V, gridded_coefs(C_D.δe)); put_at!(b, AT_CD_BETA_V, gridded_coefs(C_D.β))
    put_at!(b, AT_CD_ALPHA_K, gridded_knots(C_D.α_δf)[1]); put_at!(b, AT_CD_ALPHA_DF_V, gridded_coefs(C_D.α_δf))       # 26 x 4
    put_at!(b, AT_CY_BETA_K, gridded_knots(C_Y.β_δf)[1]); put_at!(b, AT_DF2_K, gridded_knots(C_Y.β_δf)[2]); put_at!(b, AT_CY_BETA_DF_V, gridded_coefs(C_Y.β_δf))   # 3 x 2
    @assert gridded_knots(C_Y.p) == gridded_knots(C_Y.r) == gridded_knots(C_l.r) && gridded_knots(C_Y.p)[2] == gridded_knots(C_Y.β_δf)[2]
    put_at!(b, AT_ALPHA2_K, gridded_knots(C_Y.p)[1]); put_at!(b, AT_CY_P_V, gridded_coefs(C_Y.p)); put_at!(b, AT_CY_R_V, gridded_coefs(C_Y.r)); put_at!(b, AT_CL_R_V, gridded_coefs(C_l.r))
    put_at!(b, AT_CL_ALPHA_K, gridded_knots(C_L.α)[1]); put_at!(b, AT_CL_ALPHA_V, gridded_coefs(C_L.α))                 # 17 x 2 (stall 0 / 1)
    put_at!(b, AT_SCALARS, Float64[C_D.z, C_Y.δr, C_Y.δa, C_L.δe, C_L.q, C_L.α_dot, C_l.δa, C_l.δr, C_l.β, C_l.p,          # AS_* order of csrc/tables.h
                                C_m.z, C_m.δe, C_m.α, C_m.q, C_m.α_dot, C_n.δr, C_n.δa, C_n.β, C_n.p, C_n.r])
    return b
end

# ---- the verbs --------------------------------------------------------------------------------------------
state(w::BatchedWorld) = (x = Matrix{Float64}(undef, w.n, w.nx); s = Matrix{Int32}(undef, w.n, NS);
    check(ccall((:fb_get_state, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Int32}), w.handle, x, s)); x)
outputs(w::BatchedWorld) = (y = Matrix{Float64}(undef, w.n, NY);
    check(ccall((:fb_get_outputs, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, y)); y)

"`outputs(world, mask)`: only the blocks of mdl.y named by the FB_YF_* bits (KIN = 1, AIR = 2, AERO = 4, LDG = 8, PWP = 16, FUEL = 32, DYN = 64)"
function outputs(w::BatchedWorld, mask::Integer)
    first = (0, 40, 62, 78, 111, 133, 134, NY)
    rows = sum(first[i + 1] - first[i] for i in 1:7 if mask & (1 << (i - 1)) != 0)
    y = Matrix{Float64}(undef, w.n, rows)
    check(ccall((:fb_get_output_fields, lib), Cint, (Ptr{Cvoid}, UInt32, Ptr{Cdouble}), w.handle, mask, y)); y
end

"f_init!(world, C172.TrimParameters()) — one trim per aircraft, on the device."
function f_init!(w::BatchedWorld, trim::C172.TrimParameters)
    tp = Matrix{Float64}(undef, w.n, NTP)
    tp[:, 1:3] .= trim.Ob.loc[:]'; tp[:, 4] .= Float64(trim.Ob.h); tp[:, 5] .= trim.ψ_nb; tp[:, 6] .= trim.EAS
    tp[:, 7] .= trim.γ_wb_n; tp[:, 8] .= trim.ψ_wb_dot; tp[:, 9] .= trim.θ_wb_dot; tp[:, 10] .= trim.β_a
    tp[:, 11] .= Float64(trim.fuel_load); tp[:, 12] .= Float64(trim.mixture); tp[:, 13] .= Float64(trim.flaps)
    p = trim.payload
    tp[:, 14:18] .= Float64[Float64(p.m_pilot) Float64(p.m_copilot) Float64(p.m_lpass) Float64(p.m_rpass) Float64(p.m_baggage)]
    ts = repeat(collect(C172.TrimState())', w.n)          # initial guess, c172.jl:796-804
    ok = Vector{Int32}(undef, w.n); cost = Vector{Float64}(undef, w.n)
    check(ccall((:fb_trim, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Int32}, Ptr{Cdouble}), w.handle, tp, ts, ok, cost))
    all(==(1), ok) || @warn("Trimming failed for $(count(!=(1), ok)) aircraft")    # c172.jl:936-938
    return nothing
end
f_ode!(w::BatchedWorld) = (check(ccall((:fb_f_ode, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, C_NULL)); nothing)
f_step!(w::BatchedWorld) = (check(ccall((:fb_f_step, lib), Cint, (Ptr{Cvoid},), w.handle)); nothing)
f_periodic!(w::BatchedWorld) = (check(ccall((:fb_f_periodic, lib), Cint, (Ptr{Cvoid},), w.handle)); nothing)

"Simulation(world; dt, Δt) for the batch: fixed-step RK4 + Flight.jl's callback order, fused on the GPU."
mutable struct BatchedSimulation
    mdl::BatchedWorld
    dt::Float64
    Δt::Float64
    nstep::Int
    function BatchedSimulation(mdl::BatchedWorld; dt::Real = 0.02, Δt::Real = dt, steps_per_launch::Integer = 50)
        p = Ref{Params}()
        check(ccall((:fb_get_params, lib), Cint, (Ptr{Cvoid}, Ptr{Params}), mdl.handle, p))
        q = p[]
        p[] = Params(dt, round(Cint, Δt / dt), q.surface, q.T_sl, q.p_sl, q.wind_ned, q.h_terrain)
        check(ccall((:fb_set_params, lib), Cint, (Ptr{Cvoid}, Ptr{Params}), mdl.handle, p))
        check(ccall((:fb_set_steps_per_launch, lib), Cint, (Ptr{Cvoid}, Cint), mdl.handle, steps_per_launch))
        new(mdl, dt, Δt, 0)
    end
end
"""Each simulation's own environment: env [N x 6], columns wind N / E / D (`world.atmosphere.wind.u`, FlightPhysics/src/atmosphere.jl:156-165),
sea-level T / p (`world.atmosphere.sl.u`, :75-84), terrain elevation (FlightPhysics/src/terrain.jl:34-36). `nothing` returns to the batch-wide block."""
function set_env!(w::BatchedWorld, env::Union{Matrix{Float64}, Nothing})
    # the C ABI takes no length: a matrix of another size would be read out of bounds, a transposed 6 x N one silently misread
    env === nothing || size(env) == (w.n, 6) || throw(DimensionMismatch("set_env!: env must be $(w.n) x 6 (aircraft x wind N / E / D, T_sl, p_sl, h_terrain), got $(size(env))"))
    check(ccall((:fb_set_env, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, env === nothing ? C_NULL : env))
    nothing
end
"f_init!(world) for a Cessna172Xv2 batch whose state and inputs the host has set (C172.Init-style initial condition): the avionics half of f_init!."
function f_init!(w::BatchedWorld)
    check(ccall((:fb_f_init, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Cint), w.handle, C_NULL, 0))
    nothing
end
init!(sim::BatchedSimulation, args...) = (f_init!(sim.mdl, args...); sim.nstep = 0; nothing)
function step!(sim::BatchedSimulation, Δt_total::Real = sim.dt, stop_at_tdt::Bool = true)
    n = round(Int, Δt_total / sim.dt)
    check(ccall((:fb_step, lib), Cint, (Ptr{Cvoid}, Int64), sim.mdl.handle, n))
    check(ccall((:fb_sync, lib), Cint, (Ptr{Cvoid},), sim.mdl.handle))
    sim.nstep += n
    return nothing
end
Base.getproperty(sim::BatchedSimulation, s::Symbol) = s === :t ? getfield(sim, :nstep) * getfield(sim, :dt) : getfield(sim, s)

# ---- scripted scenarios: `user_callback!` as a table on the device (include/flightbatch.h, FB_TABLE_SCENARIO) ------------------------
# Simulation(mdl; user_callback!) calls a closure after every step (FC/sim.jl:185, 334-336); for a batch the closures of the demos
# (FlightApps/demos/c172_demos.jl:423-486, 525-642: a phase symbol, per phase "set these inputs; if <condition> set those, next phase") are
# a table of phases, rules and actions that a kernel interprets between the stepping launches. `blob` is the packed table (the layout is in
# the header; flightbatch/scenario.py builds it on the Python side), `par` the per-aircraft parameter rows [N x n_par].
const TABLE_SCENARIO = Cint(6)
function set_scenario!(w::BatchedWorld, blob::Vector{Float64}, par::Union{Matrix{Float64}, Nothing} = nothing; every::Integer = 1)
    len = Ref{Int64}(length(blob))
    check(ccall((:fb_set_table, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Int64}, Cint), w.handle, TABLE_SCENARIO, blob, len, 1))
    check(ccall((:fb_scenario_configure, lib), Cint, (Ptr{Cvoid}, Cint), w.handle, every))
    n_par = round(Int, blob[5])
    if n_par > 0
        par !== nothing && size(par) == (w.n, n_par) || throw(DimensionMismatch("set_scenario!: par must be $(w.n) x $(n_par)"))
        check(ccall((:fb_scenario_set_params, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, par))
    end
    nothing
end
clear_scenario!(w::BatchedWorld) = (check(ccall((:fb_scenario_configure, lib), Cint, (Ptr{Cvoid}, Cint), w.handle, 0)); nothing)
"every aircraft's phase, the step at which it entered it, and its record rows [N x n_rec]"
function scenario_state(w::BatchedWorld, n_rec::Integer)
    phase = Vector{Int32}(undef, w.n); since = Vector{Int64}(undef, w.n); rec = Matrix{Float64}(undef, w.n, max(n_rec, 1))
    check(ccall((:fb_scenario_get_state, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int64}, Ptr{Cdouble}), w.handle, phase, since, rec))
    return phase, since, rec[:, 1:n_rec]
end

# ---- multi-GPU trajectory collection: one process per GPU, one RCCL all-gather of the state panels (include/flightbatch.h) -------
"rank 0: `id = comm_unique_id()`, hand the 128 bytes to the other ranks (file, MPI, sockets); every rank: `comm_init(world_handle, nranks, rank, id)`"
comm_unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:fb_comm_unique_id, lib), Cint, (Ptr{UInt8},), id)); id)
function comm_init(w::BatchedWorld, nranks::Integer, rank::Integer, id::Vector{UInt8})
    comm = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fb_comm_init, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ptr{Ptr{Cvoid}}), w.handle, nranks, rank, id, comm))
    return comm[]
end
"shard sizes of all ranks, exchanged by `comm_init`: `(n_of, n_max)`"
function comm_shard_sizes(comm::Ptr{Cvoid}, nranks::Integer)
    n_of = Vector{Int64}(undef, nranks); n_max = Ref{Int64}(0)
    check(ccall((:fb_comm_shard_sizes, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), comm, n_of, n_max))
    return n_of, n_max[]
end
"all-gather of the device-layout state into `recv_dev` (device pointer to nranks x Nx x n_max doubles; equal shards: n_max = N), asynchronous on the world's stream"
gather_state!(w::BatchedWorld, comm::Ptr{Cvoid}, recv_dev::Ptr{Cvoid}) =
    check(ccall((:fb_gather_state, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), w.handle, comm, recv_dev))

# ---- saving: the SavingCallback / TimeSeries(sim) of FC/sim.jl:210-217,644-704, kept on the device -------------------
const LOG_X0 = Cint(1000)   # FB_LOG_X0: rows >= LOG_X0 select state rows, rows < LOG_X0 rows of the output record y
"Log `rows` (0-based, see include/flightbatch.h FB_Y_*) every `saveat` seconds into a device buffer of `capacity` samples."
function save_on!(sim::BatchedSimulation, rows::Vector{Cint}; saveat::Real = sim.dt, capacity::Integer = 1024)
    check(ccall((:fb_log_configure, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cint}, Cint),
                sim.mdl.handle, round(Int64, saveat / sim.dt), capacity, rows, length(rows)))
    check(ccall((:fb_log_record, lib), Cint, (Ptr{Cvoid},), sim.mdl.handle))   # y(t0), like reinit! does
end
"TimeSeries(sim) for the batch: (t, data) with data[aircraft, row, sample]."
function timeseries(sim::BatchedSimulation, nrows::Integer)
    cnt = Ref{Int64}(0)
    check(ccall((:fb_log_count, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), sim.mdl.handle, cnt))
    t = Vector{Float64}(undef, cnt[])
    data = Array{Float64, 3}(undef, sim.mdl.n, nrows, cnt[])
    check(ccall((:fb_log_read, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}), sim.mdl.handle, 0, cnt[], t, data))
    return t, data
end

end # module
