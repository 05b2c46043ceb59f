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
using Flight.FlightApps: C172
using Flight.FlightPhysics: Propellers, Piston, Geodesy

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
`aircraft` is `Cessna172Sv0(kin)` (kin = WA(), ECEF() or NED()) or `Cessna172Xv2()`; its type picks the model id."
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
        w = new(h[], n, nx[])      # 27 (WA) / 26 (ECEF) / 24 (NED) / 34 (Xv2): length(Model(aircraft).x)
        finalizer(w -> ccall((:fb_destroy, lib), Cint, (Ptr{Cvoid},), w.handle), w)
        upload_tables!(w)
        # Xv2: the ten gain lookups of c172x/control/data packed as include/flightbatch.h documents (flightbatch/ctl_gains.py)
        aircraft isa Cessna172Xv2 && set_table!(w, TABLE_CTL_GAINS, pack_ctl_gains(aircraft.avionics.ctl))
        return w
    end
end
pack_ctl_gains(ctl)::Vector{Float64} = error("see flightbatch/ctl_gains.py: ctl_gains_blob() for the layout to replicate")

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
    read!(joinpath(dirname(pathof(Geodesy.eval(:(@__MODULE__)))), "data", "ww15mgh_le.bin"), egm)
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

# The two packers read the interpolation objects' knots/coefs and lay them out as csrc/tables.h documents
# (AT_* / PT_* offsets); flightbatch/tables.py is the executed equivalent and serves as their specification.
pack_piston(lookup)::Vector{Float64} = error("see flightbatch/tables.py: piston_blob() for the layout to replicate")
pack_aero(lookup)::Vector{Float64} = error("see flightbatch/tables.py: aero_blob() for the layout to replicate")

# ---- the verbs --------------------------------------------------------------------------------------------
state(w::BatchedWorld) = (x = Matrix{Float64}(undef, w.n, w.nx); s = Matrix{Int32}(undef, w.n, NS);
    check(ccall((:fb_get_state, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Int32}), w.handle, x, s)); x)
outputs(w::BatchedWorld) = (y = Matrix{Float64}(undef, w.n, NY);
    check(ccall((:fb_get_outputs, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), w.handle, y)); y)

"f_init!(world, C172.TrimParameters()) — one trim per aircraft, on the device."
function f_init!(w::BatchedWorld, trim::C172.TrimParameters)
    tp = Matrix{Float64}(undef, w.n, NTP)
    tp[:, 1:3] .= trim.Ob.loc[:]'; tp[:, 4] .= Float64(trim.Ob.h); tp[:, 5] .= trim.ψ_nb; tp[:, 6] .= trim.EAS
    tp[:, 7] .= trim.γ_wb_n; tp[:, 8] .= trim.ψ_wb_dot; tp[:, 9] .= trim.θ_wb_dot; tp[:, 10] .= trim.β_a
    tp[:, 11] .= Float64(trim.fuel_load); tp[:, 12] .= Float64(trim.mixture); tp[:, 13] .= Float64(trim.flaps)
    p = trim.payload
    tp[:, 14:18] .= Float64[p.m_pilot p.m_copilot p.m_lpass p.m_rpass p.m_baggage]
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

# ---- multi-GPU trajectory collection: one process per GPU, one RCCL all-gather of the state panels (include/flightbatch.h) -------
"rank 0: `id = comm_unique_id()`, hand the 128 bytes to the other ranks (file, MPI, sockets); every rank: `comm_init(world_handle, nranks, rank, id)`"
comm_unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:fb_comm_unique_id, lib), Cint, (Ptr{UInt8},), id)); id)
function comm_init(w::BatchedWorld, nranks::Integer, rank::Integer, id::Vector{UInt8})
    comm = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:fb_comm_init, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ptr{Ptr{Cvoid}}), w.handle, nranks, rank, id, comm))
    return comm[]
end
"all-gather of the device-layout state into `recv_dev` (device pointer to nranks x Nx x N doubles), asynchronous on the world's stream"
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
