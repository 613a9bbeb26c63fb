# TrajOptHIP.jl — the Julia side of the drop-in boundary: binds every entry point of include/trajopt_hip.h with `ccall`
# and hangs them on TrajectoryOptimization.jl's own verbs (src/TrajectoryOptimization.jl:29-71), so that code written
# against the reference keeps its names while a BATCH of problems runs on one MI355X behind `libtrajopt_hip.so`.
#
# Julia is not installed in the build image of this repository: the file is written against Julia 1.x / the v0.7.1
# sources of the reference and cannot be executed there.  tests/test_julia_shim.py checks it structurally: every `to_*`
# symbol of the header appears in a `ccall` here with the header's argument count, and every struct mirrors the header's
# field list.
module TrajOptHIP

using LinearAlgebra
using StaticArrays
import RobotDynamics
import RobotZoo
import Rotations
import TrajectoryOptimization
const RD = RobotDynamics
const TO = TrajectoryOptimization

const lib = get(ENV, "TRAJOPT_HIP_LIBRARY", "libtrajopt_hip")   # trajectoryoptimization.jl_amd/csrc/libtrajopt_hip.so

# ------------------------------------------------------------------------------------------------ header mirrors
const TO_ABI_VERSION = Int32(6)
const MAXN, MAXM, MAXP, MAXPAR, MAXIND = 16, 8, 40, 400, 48
const PROFILE_SLOTS = 4

struct CostDesc                      # == to_cost_desc
    kind::Int32
    terminal::Int32
    Q::NTuple{MAXN * MAXN,Float64}
    R::NTuple{MAXM * MAXM,Float64}
    H::NTuple{MAXM * MAXN,Float64}
    q::NTuple{MAXN,Float64}
    r::NTuple{MAXM,Float64}
    c::Float64
    w::Float64
    q_ref::NTuple{4,Float64}
    q_ind::NTuple{4,Int32}
end

struct ConstraintDesc                # == to_constraint_desc
    kind::Int32
    sense::Int32
    k_first::Int32
    k_last::Int32
    p::Int32
    n_inds::Int32
    inds::NTuple{MAXIND,Int32}
    n_params::Int32
    params::NTuple{MAXPAR,Float64}
end

struct StepModel                     # == to_step_model (one time step of a model vector, TO_MODEL_VECTOR)
    kind::Int32                      # 0 double integrator, 1 Cartpole, 2 linear map
    n::Int32
    m::Int32
    n_out::Int32
    params::NTuple{60,Float64}
end

struct ProblemDesc                   # == to_problem_desc
    abi_version::Int32
    model::Int32
    integrator::Int32
    n::Int32
    m::Int32
    N::Int32
    B::Int32
    model_params::NTuple{16,Float64}
    t0::Float64
    tf::Float64
    dt::Ptr{Float64}
    n_costs::Int32
    costs::Ptr{CostDesc}
    cost_index::Ptr{Int32}
    n_constraints::Int32
    constraints::Ptr{ConstraintDesc}
    step_models::Ptr{StepModel}          # TO_MODEL_VECTOR: one model per time step (N-1), C_NULL otherwise
end

mutable struct SolverOpts            # == to_solver_opts (names of Altro.SolverOptions)
    cost_tolerance::Float64
    gradient_tolerance::Float64
    iterations::Int32
    dJ_counter_limit::Int32
    iterations_linesearch::Int32
    reserved0::Int32
    line_search_lower_bound::Float64
    line_search_upper_bound::Float64
    line_search_decrease_factor::Float64
    bp_reg_initial::Float64
    bp_reg_increase_factor::Float64
    bp_reg_min::Float64
    bp_reg_max::Float64
    bp_reg_fp::Float64
    max_cost_value::Float64
    max_state_value::Float64
    max_control_value::Float64
    constraint_tolerance::Float64
    cost_tolerance_intermediate::Float64
    penalty_initial::Float64
    penalty_scaling::Float64
    penalty_max::Float64
    dual_max::Float64
    iterations_outer::Int32
    cost_dt_scaling::Int32
    iterations_total::Int32
    al_full_newton::Int32
    # projected-Newton polish (Altro.ProjectedNewtonSolver; names of Altro.SolverOptions, ρ spelled out)
    projected_newton_tolerance::Float64
    active_set_tolerance_pn::Float64
    rho_chol::Float64
    rho_primal::Float64
    r_threshold::Float64
    n_steps::Int32
    projected_newton::Int32
    SolverOpts() = new()
end

mutable struct SolveStats            # == to_solve_stats
    iterations::Ptr{Int32}
    iterations_outer::Ptr{Int32}
    status::Ptr{Int32}
    cost::Ptr{Float64}
    dJ::Ptr{Float64}
    gradient::Ptr{Float64}
    c_max::Ptr{Float64}
    penalty_max::Ptr{Float64}
    iterations_pn::Ptr{Int32}
    total_iterations::Int64
    batch_steps::Int32
    reserved::Int32
    solve_ms::Float64
end

@enum SolverStatus::Int32 UNSOLVED = 0 LINESEARCH_FAIL SOLVE_SUCCEEDED MAX_ITERATIONS MAX_ITERATIONS_OUTER MAXIMUM_COST STATE_LIMIT CONTROL_LIMIT NO_PROGRESS COST_INCREASE REGULARIZATION_MAX PROJECTION_FAIL

# ------------------------------------------------------------------------------------------------ errors
last_error() = unsafe_string(ccall((:to_last_error, lib), Cstring, ()))
abi_version() = ccall((:to_abi_version, lib), Cint, ())
build_id() = unsafe_string(ccall((:to_build_id, lib), Cstring, ()))
"A library of another ABI version is refused at load time (include/trajopt_hip.h: the version changes with every new symbol or field)."
function __init__()
    v = abi_version()
    v == TO_ABI_VERSION || error("libtrajopt_hip.so has ABI version $v, this shim was written for $TO_ABI_VERSION: rebuild one of them")
end

"Negative return codes become the exception the reference throws (include/trajopt_hip.h, to_status_code)."
function check(rc::Integer)
    rc >= 0 && return rc
    msg = last_error()
    rc == -1 && throw(DimensionMismatch(msg))      # src/problem.jl:64-68, src/constraint_list.jl:109
    rc == -2 && throw(ArgumentError(msg))          # src/problem.jl:88, src/constraints.jl:712
    rc == -3 && throw(AssertionError(msg))         # src/problem.jl:49-55
    rc == -7 && error(msg)                         # "Invalid second-order cone projection" src/cones.jl:124
    error("libtrajopt_hip: $msg (code $rc)")
end

function device_count()
    n = Ref{Cint}(0)
    check(ccall((:to_device_count, lib), Cint, (Ref{Cint},), n))
    Int(n[])
end

function default_options()
    o = SolverOpts()
    check(ccall((:to_default_options, lib), Cint, (Ref{SolverOpts},), o))
    o
end

"`SolverOpts(; cost_tolerance = 1e-4, penalty_scaling = 10.0, ...)`: the defaults with the given fields replaced."
function solver_options(; kwargs...)
    o = default_options()
    for (k, v) in kwargs
        setfield!(o, k, convert(fieldtype(SolverOpts, k), v))
    end
    o
end

# ------------------------------------------------------------------------------------------------ descriptors
pad(v, n) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, n)
padi(v, n) = ntuple(i -> i <= length(v) ? Int32(v[i]) : Int32(0), n)
const NOQUAT = ((1.0, 0.0, 0.0, 0.0), Int32.((4, 5, 6, 7)))

# cost functions (src/cost_functions.jl:326-453, src/lie_costs.jl:34-55, 178-241)
costdesc(c::TO.DiagonalCost) = CostDesc(0, c.terminal, pad(diag(c.Q), MAXN * MAXN), pad(diag(c.R), MAXM * MAXM),
    pad(Float64[], MAXM * MAXN), pad(c.q, MAXN), pad(c.r, MAXM), c.c, 0.0, NOQUAT...)
costdesc(c::TO.QuadraticCost) = CostDesc(1, c.terminal, pad(vec(Matrix(c.Q)), MAXN * MAXN), pad(vec(Matrix(c.R)), MAXM * MAXM),
    pad(vec(Matrix(c.H)), MAXM * MAXN), pad(c.q, MAXN), pad(c.r, MAXM), c.c, 0.0, NOQUAT...)
costdesc(c::TO.DiagonalQuatCost) = CostDesc(2, c.terminal, pad(diag(c.Q), MAXN * MAXN), pad(diag(c.R), MAXM * MAXM),
    pad(Float64[], MAXM * MAXN), pad(c.q, MAXN), pad(c.r, MAXM), c.c, c.w, Tuple(Float64.(c.q_ref)), Int32.(Tuple(c.q_ind)))
# ErrorQuadratic{Rot}: w carries to_rotation of the model's state (0 QuatRotation, 1 MRP, 2 RodriguesParam; src/lie_costs.jl:1-3,178-241)
rotationcode(::Type{<:Rotations.QuatRotation}) = 0.0
rotationcode(::Type{<:Rotations.MRP}) = 1.0
rotationcode(::Type{<:Rotations.RodriguesParam}) = 2.0
costdesc(c::TO.ErrorQuadratic{Rot}) where {Rot} = CostDesc(3, false, pad(diag(c.Q), MAXN * MAXN), pad(diag(c.R), MAXM * MAXM),
    pad(Float64[], MAXM * MAXN), pad(c.x_ref, MAXN), pad(c.r, MAXM), c.c, rotationcode(Rot), (1.0, 0.0, 0.0, 0.0), Int32.(Tuple(c.q_ind)))

# constraints (src/constraints.jl); sense codes = to_cone
sensecode(::TO.Equality) = Int32(0)
sensecode(::TO.Inequality) = Int32(1)
sensecode(::TO.SecondOrderCone) = Int32(2)
conecode(::TO.ZeroCone) = Int32(0)
conecode(::TO.NegativeOrthant) = Int32(1)
conecode(::TO.SecondOrderCone) = Int32(2)
conecode(::TO.PositiveOrthant) = Int32(3)
conecode(::TO.IdentityCone) = Int32(4)

_con(kind, sense, r, inds, params) = ConstraintDesc(kind, sense, first(r), last(r), 0,
    length(inds), padi(inds, MAXIND), length(params), pad(params, MAXPAR))

condesc(con::TO.GoalConstraint, r) = _con(0, 0, r, con.inds, con.xf)                                   # :22-87
condesc(con::TO.BoundConstraint, r) = _con(1, 1, r, Int[], [con.z_max; con.z_min])                      # :644-783
condesc(con::TO.NormConstraint, r) = _con(2, sensecode(con.sense), r, con.inds, [con.val])              # :438-521
condesc(con::TO.CircleConstraint, r) = _con(3, 1, r, [con.xi, con.yi], [con.x; con.y; con.radius])      # :168-233
condesc(con::TO.SphereConstraint, r) = _con(4, 1, r, [con.xi, con.yi, con.zi], [con.x; con.y; con.z; con.radius])  # :249-326
condesc(con::TO.LinearConstraint, r) = _con(5, sensecode(con.sense), r, con.inds, [vec(Matrix(con.A)); con.b])     # :103-150
condesc(con::TO.CollisionConstraint, r) = _con(6, 1, r, [con.x1; con.x2], [con.radius])                 # :332-393
condesc(con::TO.QuatVecEq, r) = _con(7, 0, r, con.qind, Rotations.params(con.qf))                       # :938-965
function condesc(con::TO.StateBound, r)                                                                 # :528-631
    n, m = RD.state_dim(con), RD.control_dim(con)
    _con(1, 1, r, Int[], [con.z_max; fill(Inf, m); con.z_min; fill(-Inf, m)])
end
function condesc(con::TO.ControlBound, r)
    n, m = RD.state_dim(con), RD.control_dim(con)
    _con(1, 1, r, Int[], [fill(Inf, n); con.z_max; fill(-Inf, n); con.z_min])
end

"IndexedConstraint (src/constraints.jl:820-936) needs no kernel: the inner descriptor's indices are moved to the slice."
function condesc(con::TO.IndexedConstraint, r)
    d = condesc(con.con, r)
    n, m, n0, m0 = con.n, con.m, con.n0, con.m0
    newidx(i) = i <= n0 ? con.ix[i] : con.iu[i - n0]          # con.iu is already offset by n (:853)
    if d.kind == 1                                             # bounds: params = [z_max; z_min] of the inner dimensions
        zmax, zmin = fill(Inf, n + m), fill(-Inf, n + m)
        for i in 1:(n0 + m0)
            zmax[newidx(i)] = d.params[i]
            zmin[newidx(i)] = d.params[n0 + m0 + i]
        end
        return _con(1, 1, r, Int[], [zmax; zmin])
    end
    inds = [newidx(Int(d.inds[i])) for i in 1:d.n_inds]
    ConstraintDesc(d.kind, d.sense, d.k_first, d.k_last, d.p, d.n_inds, padi(inds, MAXIND), d.n_params, d.params)
end

# models (to_model_id): parameters in the order documented in the header
modelid(::RobotZoo.Cartpole) = Int32(1)
modelid(::RobotZoo.Quadrotor) = Int32(2)
modelid(::RobotZoo.DoubleIntegrator) = Int32(0)
modelparams(c::RobotZoo.Cartpole) = pad([c.mc, c.mp, c.l, c.g], 16)
modelparams(q::RobotZoo.Quadrotor{R}) where {R} = pad([q.mass, q.J[1, 1], q.J[2, 2], q.J[3, 3], q.gravity..., q.motor_dist, q.kf, q.km,
    rotationcode(R)], 16)   # params[10] = attitude representation of the state: Quadrotor{QuatRotation} n = 13, {MRP} / {RodriguesParam} n = 12
modelparams(d::RobotZoo.DoubleIntegrator{N,M}) where {N,M} = pad([1.0, M], 16)   # RobotZoo's double integrator is ẍ = u: unit mass
# a user-defined double integrator with a `mass` field, like the one examples/quickstart.jl:11-23 defines (n = 2D, m = D)
struct MassDoubleIntegrator{D} <: RD.ContinuousDynamics
    mass::Float64
end
RD.state_dim(::MassDoubleIntegrator{D}) where {D} = 2D
RD.control_dim(::MassDoubleIntegrator{D}) where {D} = D
RD.dynamics(model::MassDoubleIntegrator{D}, x, u) where {D} = [x[D+1:2D]; u ./ model.mass]
modelid(::MassDoubleIntegrator) = Int32(0)
modelparams(d::MassDoubleIntegrator{D}) where {D} = pad([d.mass, D], 16)
# Altro's InfeasibleModel (the state augmentation of ALTRO's infeasible start, TO_MODEL_INFEASIBLE): x⁺ = f_d(x, u[1:m0]) + u[m0+1:end].
# Altro builds it (and the lifted costs / constraints: TO.change_dimension, src/constraints.jl:820-936, src/constraint_list.jl:208-217,
# src/cost_functions.jl:391-401, plus its InfeasibleConstraint) in ALTROSolver(prob; infeasible=true); here the wrapper only has to name
# the base model.  `InfeasibleBase(model)` stands in for Altro.InfeasibleModel when Altro is not loaded.
struct InfeasibleBase{M<:RD.ContinuousDynamics} <: RD.ContinuousDynamics
    model::M
end
RD.state_dim(m::InfeasibleBase) = RD.state_dim(m.model)
RD.control_dim(m::InfeasibleBase) = RD.control_dim(m.model) + RD.state_dim(m.model)
modelid(::InfeasibleBase) = Int32(5)
function modelparams(m::InfeasibleBase)
    p = modelparams(m.model)
    p[16] = Float64(modelid(m.model))      # model_params[15] (0-based): to_model_id of the base
    p
end

# ---- model vectors (Problem(models::Vector{<:DiscreteDynamics}, ...), src/problem.jl:36-73, src/dynamics.jl:15-31): TO_MODEL_VECTOR
"""
    LinearMap(A, B)

The discrete map x⁺ = A x + B u from (size(A,2), size(B,2)) to size(A,1) states: the kind of jump map that connects the phases
of a model vector (test/hybrid_dynamics_model.jl:31-33 is one).
"""
struct LinearMap <: RD.DiscreteDynamics
    A::Matrix{Float64}
    B::Matrix{Float64}
end
RD.state_dim(f::LinearMap) = size(f.A, 2)
RD.control_dim(f::LinearMap) = size(f.B, 2)
RD.output_dim(f::LinearMap) = size(f.A, 1)
RD.discrete_dynamics(f::LinearMap, x, u, t, dt) = f.A * x + f.B * u
stepmodel(f::LinearMap) = StepModel(2, size(f.A, 2), size(f.B, 2), size(f.A, 1), pad([vec(f.A); vec(f.B)], 60))
stepmodel(d::RD.DiscretizedDynamics) = stepmodel(continuous(d))
stepmodel(c::RobotZoo.Cartpole) = StepModel(1, 4, 1, 4, pad([c.mc, c.mp, c.l, c.g], 60))
stepmodel(d::RobotZoo.DoubleIntegrator{N,M}) where {N,M} = StepModel(0, 2M, M, 2M, pad([1.0], 60))
stepmodel(d::MassDoubleIntegrator{D}) where {D} = StepModel(0, 2D, D, 2D, pad([d.mass], 60))
const VECTOR_N, VECTOR_M = 6, 3   # TO_VECTOR_N, TO_VECTOR_M: the storage dimensions of a model vector
"A knot's cost on the zero-padded (6, 3) vectors: nothing on padded states; a padded control gets R = 1, so it stays exactly 0."
function padcost(d::CostDesc, m0::Integer)
    R = collect(d.R)
    d.kind == 1 && error("model vector: give a dense QuadraticCost at the padded dimensions (6, 3) yourself (Q, R, H are stored column-major at their own size)")
    for j in m0+1:VECTOR_M
        R[j] = 1.0
    end
    CostDesc(d.kind, d.terminal, d.Q, Tuple(R), d.H, d.q, d.r, d.c, d.w, d.q_ref, d.q_ind)
end
"One model for every step?  (Problem(model, ...) builds N-1 copies of one model, src/problem.jl:115.)"
uniform_models(models) = all(m -> typeof(m) == typeof(models[1]) && m == models[1], models)
continuous(model::RD.DiscretizedDynamics) = model.continuous_dynamics
integratorid(::RD.DiscretizedDynamics{<:Any,<:RD.RK4}) = Int32(0)
integratorid(::RD.DiscretizedDynamics{<:Any,<:RD.RK3}) = Int32(1)
integratorid(::RD.DiscretizedDynamics{<:Any,<:RD.Euler}) = Int32(2)

# ------------------------------------------------------------------------------------------------ BatchProblem
"A `TO.Problem` replicated over `B` independent trajectories on one GPU (trajectory `b` has its own x0, U0, duals)."
mutable struct BatchProblem
    prob::TO.Problem
    handle::Ptr{Cvoid}
    B::Int
    n::Int
    m::Int
    ne::Int
    N::Int
end

function BatchProblem(prob::TO.Problem, B::Integer; device::Integer = 0, opts::Union{Nothing,SolverOpts} = nothing)
    n, m, N = RD.dims(prob)
    # distinct cost OBJECTS (identity, not value: set_LQR_goal! mutates them in place), in order of first appearance, and for
    # every knot the 0-based position of its cost in that list (to_problem_desc.cost_index; Objective.cost src/objective.jl:28)
    costs = Any[]
    for c in prob.obj.cost
        any(x -> x === c, costs) || push!(costs, c)
    end
    descs = CostDesc[costdesc(c) for c in costs]
    index = Int32[findfirst(c -> c === prob.obj.cost[k], costs) - 1 for k in 1:N]
    cons = ConstraintDesc[condesc(con, inds) for (inds, con) in zip(prob.constraints)]   # Base.zip(::ConstraintList) = zip(inds, constraints), src/constraint_list.jl:147
    dts = Float64[prob.Z[k].dt for k in 1:N-1]
    model = prob.model[1]
    steps = StepModel[]
    if uniform_models(prob.model)
        desc = ProblemDesc(TO_ABI_VERSION, modelid(continuous(model)), integratorid(model), n, m, N, B,
            modelparams(continuous(model)), TO.get_initial_time(prob), TO.get_final_time(prob),   # src/problem.jl:186-196 (Problem has no tf field)
            pointer(dts), length(descs), pointer(descs), pointer(index),
            length(cons), isempty(cons) ? Ptr{ConstraintDesc}(C_NULL) : pointer(cons), Ptr{StepModel}(C_NULL))
    else
        # a model vector: one table entry per time step; the library stores the trajectory at (6, 3) with the narrower knots
        # zero-padded, so the per-knot costs / constraints must be given at (6, 3) as well (pad them like the Python mirror's
        # pad_cost / IndexedConstraint: nothing on padded states, R = 1 on padded controls)
        steps = StepModel[stepmodel(f) for f in prob.model]
        descs = CostDesc[padcost(costdesc(c), RD.control_dim(c)) for c in costs]
        # constraints of the narrower knots: wrap them in TO.IndexedConstraint(VECTOR_N, VECTOR_M, con, 1:n0, 1:m0) before building
        # the Problem (condesc lowers the wrapper by remapping the inner constraint's indices onto the padded [x; u])
        first_discretized = findfirst(f -> f isa RD.DiscretizedDynamics, prob.model)
        integ = first_discretized === nothing ? Int32(0) : integratorid(prob.model[first_discretized])
        desc = ProblemDesc(TO_ABI_VERSION, Int32(4), integ, VECTOR_N, VECTOR_M, N, B,
            pad(Float64[], 16), TO.get_initial_time(prob), TO.get_final_time(prob),
            pointer(dts), length(descs), pointer(descs), pointer(index),
            length(cons), isempty(cons) ? Ptr{ConstraintDesc}(C_NULL) : pointer(cons), pointer(steps))
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve descs index cons dts steps begin
        if opts === nothing
            check(ccall((:to_create, lib), Cint, (Ref{ProblemDesc}, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), desc, C_NULL, device, h))
        else
            check(ccall((:to_create, lib), Cint, (Ref{ProblemDesc}, Ref{SolverOpts}, Cint, Ref{Ptr{Cvoid}}), desc, opts, device, h))
        end
    end
    dn, dm, dne, dN, dB = Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0)
    check(ccall((:to_dims, lib), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}), h[], dn, dm, dne, dN, dB))
    bp = BatchProblem(prob, h[], B, dn[], dm[], dne[], dN[])
    finalizer(p -> ccall((:to_destroy, lib), Cint, (Ptr{Cvoid},), p.handle), bp)
    TO.set_initial_state!(bp, repeat(Vector(prob.x0), 1, B))
    TO.initial_controls!(bp, repeat(hcat(Vector.(TO.controls(prob))...), 1, 1, B))
    bp
end

set_options!(p::BatchProblem, o::SolverOpts) = check(ccall((:to_set_options, lib), Cint, (Ptr{Cvoid}, Ref{SolverOpts}), p.handle, o))
function get_options(p::BatchProblem)
    o = SolverOpts()
    check(ccall((:to_get_options, lib), Cint, (Ptr{Cvoid}, Ref{SolverOpts}), p.handle, o))
    o
end
sync(p::BatchProblem) = check(ccall((:to_sync, lib), Cint, (Ptr{Cvoid},), p.handle))
stream(p::BatchProblem) = ccall((:to_stream, lib), Ptr{Cvoid}, (Ptr{Cvoid},), p.handle)   # hipStream_t

# getters of the reference pass through to the wrapped problem (src/problem.jl:198-231)
for f in (:get_model, :get_objective, :get_constraints, :get_trajectory, :get_initial_state, :get_final_state, :gettimes, :horizonlength)
    @eval TO.$f(p::BatchProblem) = TO.$f(p.prob)
end
RD.dims(p::BatchProblem) = (p.n, p.m, p.N)

"num_constraints (src/constraint_list.jl:198): constraint rows per knot point."
function TO.num_constraints(p::BatchProblem)
    v = zeros(Int32, p.N)
    check(ccall((:to_num_constraints, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}), p.handle, v))
    Int.(v)
end

# ---- trajectory I/O: arrays (n, N, B) / (m, N-1, B) in Julia's native column-major layout
TO.set_initial_state!(p::BatchProblem, x0::Matrix{Float64}) = check(ccall((:to_set_initial_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, x0))
TO.initial_controls!(p::BatchProblem, U::Array{Float64,3}) = check(ccall((:to_set_controls, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, U))
TO.initial_controls!(p::BatchProblem, u::AbstractVector) = check(ccall((:to_set_controls_uniform, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, Vector{Float64}(u)))
TO.initial_states!(p::BatchProblem, X::Array{Float64,3}) = check(ccall((:to_set_states, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, X))
function TO.states(p::BatchProblem)
    X = zeros(p.n, p.N, p.B)
    check(ccall((:to_get_states, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, X))
    X
end
function TO.controls(p::BatchProblem)
    U = zeros(p.m, p.N - 1, p.B)
    check(ccall((:to_get_controls, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, U))
    U
end
function initial_states(p::BatchProblem)
    x0 = zeros(p.n, p.B)
    check(ccall((:to_get_initial_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, x0))
    x0
end
"Device-to-device copies into caller-owned device buffers (e.g. `ROCArray` pointers), host layout."
states_device!(p::BatchProblem, dX::Ptr{Cvoid}) = check(ccall((:to_get_states_device, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), p.handle, dX))
controls_device!(p::BatchProblem, dU::Ptr{Cvoid}) = check(ccall((:to_get_controls_device, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), p.handle, dU))

# ---- goal / reference updates between solves
"set_goal_state! (src/problem.jl:294-310): new LQR goal in the costs (`set_LQR_goal!` src/cost_functions.jl:249-258) and the GoalConstraint."
function TO.set_goal_state!(p::BatchProblem, xf::AbstractVector; objective = true, constraint = true)
    TO.set_goal_state!(p.prob, xf; objective = objective, constraint = constraint)
    objective && refresh_costs!(p)
    if constraint
        for (i, (inds, con)) in enumerate(zip(p.prob.constraints))
            con isa TO.GoalConstraint || continue
            d = condesc(con, inds)
            check(ccall((:to_set_constraint, lib), Cint, (Ptr{Cvoid}, Int32, Ref{ConstraintDesc}), p.handle, i - 1, d))
        end
    end
    p
end
"update_trajectory! (src/objective.jl:198-212): the tracking objective follows a new reference window."
function TO.update_trajectory!(p::BatchProblem, Z, start = 1)
    TO.update_trajectory!(p.prob.obj, Z, start)
    refresh_costs!(p)
end
function refresh_costs!(p::BatchProblem)
    for (i, c) in enumerate(unique(p.prob.obj.cost))
        check(ccall((:to_set_cost, lib), Cint, (Ptr{Cvoid}, Int32, Ref{CostDesc}), p.handle, i - 1, costdesc(c)))
    end
    p
end

# ---- the hot path, phase by phase
TO.rollout!(p::BatchProblem) = (check(ccall((:to_rollout, lib), Cint, (Ptr{Cvoid},), p.handle)); p)   # src/problem.jl:330-340
function TO.cost(p::BatchProblem)                                                                          # src/objective.jl:89-93
    J = zeros(p.B)
    check(ccall((:to_cost, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, J))
    J
end
function stage_costs(p::BatchProblem)                                                                      # Objective.J, src/objective.jl:104-106
    Jk = zeros(p.N, p.B)
    check(ccall((:to_stage_costs, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, Jk))
    Jk
end
function al_cost(p::BatchProblem)
    J = zeros(p.B)
    check(ccall((:to_al_cost, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, J))
    J
end
expand!(p::BatchProblem) = (check(ccall((:to_expand, lib), Cint, (Ptr{Cvoid},), p.handle)); p)
backwardpass!(p::BatchProblem) = (check(ccall((:to_backward, lib), Cint, (Ptr{Cvoid},), p.handle)); p)
function forwardpass!(p::BatchProblem)
    ls, J = zeros(Int32, p.B), zeros(p.B)
    check(ccall((:to_forward, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}), p.handle, ls, J))
    (ls_index = ls, cost = J)
end

# which == :ilqr | :al | :pn | :altro; async: enqueue on the handle's worker thread and return the buffers (call `wait_solve!`)
const _SOLVE_SYMBOL = Dict(:ilqr => :to_ilqr_solve, :al => :to_al_solve, :pn => :to_pn_solve, :altro => :to_altro_solve)
function _solve!(p::BatchProblem, which::Symbol)
    its, outer, st, ipn = zeros(Int32, p.B), zeros(Int32, p.B), zeros(Int32, p.B), zeros(Int32, p.B)
    J, dJ, grad, cmax, pen = zeros(p.B), zeros(p.B), zeros(p.B), zeros(p.B), zeros(p.B)
    stats = SolveStats(pointer(its), pointer(outer), pointer(st), pointer(J), pointer(dJ), pointer(grad), pointer(cmax), pointer(pen), pointer(ipn), 0, 0, 0, 0.0)
    GC.@preserve its outer st ipn J dJ grad cmax pen begin
        if which == :ilqr
            check(ccall((:to_ilqr_solve, lib), Cint, (Ptr{Cvoid}, Ref{SolveStats}), p.handle, stats))
        elseif which == :al
            check(ccall((:to_al_solve, lib), Cint, (Ptr{Cvoid}, Ref{SolveStats}), p.handle, stats))
        elseif which == :pn
            check(ccall((:to_pn_solve, lib), Cint, (Ptr{Cvoid}, Ref{SolveStats}), p.handle, stats))
        else
            check(ccall((:to_altro_solve, lib), Cint, (Ptr{Cvoid}, Ref{SolveStats}), p.handle, stats))
        end
    end
    (iterations = its, iterations_outer = outer, iterations_pn = ipn, status = SolverStatus.(st), cost = J, dJ = dJ, gradient = grad, c_max = cmax,
     penalty_max = pen, total_iterations = stats.total_iterations, batch_steps = stats.batch_steps, solve_ms = stats.solve_ms)
end
"Altro.iLQRSolver(prob, opts) |> solve!   for the whole batch."
solve_ilqr!(p::BatchProblem) = _solve!(p, :ilqr)
"The augmented-Lagrangian stage of Altro.ALTROSolver alone, run to `constraint_tolerance`."
solve_al!(p::BatchProblem) = _solve!(p, :al)
"Altro.ProjectedNewtonSolver(prob, opts) |> solve!   on the problem's current trajectories."
solve_pn!(p::BatchProblem) = _solve!(p, :pn)
"Altro.ALTROSolver(prob, opts) |> solve!   (AL-iLQR to `projected_newton_tolerance`, then the projected-Newton polish)."
solve_altro!(p::BatchProblem) = _solve!(p, :altro)

"""
Asynchronous solves (`to_*_solve_async`): the solve runs on a worker thread owned by the handle, `wait_solve!` blocks until it is
done and returns the statistics.  One solve in flight per `BatchProblem`; two problems in flight overlap on the device.
"""
mutable struct PendingSolve
    p::BatchProblem
    stats::Base.RefValue{SolveStats}
    bufs::NamedTuple
end
function solve_async!(p::BatchProblem, which::Symbol = :altro)
    bufs = (iterations = zeros(Int32, p.B), iterations_outer = zeros(Int32, p.B), status = zeros(Int32, p.B), iterations_pn = zeros(Int32, p.B),
            cost = zeros(p.B), dJ = zeros(p.B), gradient = zeros(p.B), c_max = zeros(p.B), penalty_max = zeros(p.B))
    stats = Ref(SolveStats(pointer(bufs.iterations), pointer(bufs.iterations_outer), pointer(bufs.status), pointer(bufs.cost), pointer(bufs.dJ),
                           pointer(bufs.gradient), pointer(bufs.c_max), pointer(bufs.penalty_max), pointer(bufs.iterations_pn), 0, 0, 0, 0.0))
    if which == :ilqr
        check(ccall((:to_ilqr_solve_async, lib), Cint, (Ptr{Cvoid}, Ref{SolveStats}), p.handle, stats))
    elseif which == :al
        check(ccall((:to_al_solve_async, lib), Cint, (Ptr{Cvoid}, Ref{SolveStats}), p.handle, stats))
    else
        check(ccall((:to_altro_solve_async, lib), Cint, (Ptr{Cvoid}, Ref{SolveStats}), p.handle, stats))
    end
    PendingSolve(p, stats, bufs)   # keeps the buffers and the stats block alive until wait_solve!
end
function wait_solve!(s::PendingSolve)
    check(ccall((:to_solve_wait, lib), Cint, (Ptr{Cvoid},), s.p.handle))
    st = s.stats[]
    merge(s.bufs, (status = SolverStatus.(s.bufs.status), total_iterations = st.total_iterations, batch_steps = st.batch_steps, solve_ms = st.solve_ms))
end

"""
    solve_progress(s::PendingSolve) -> (active, batch_steps, in_flight)
    wait_below!(s::PendingSolve, active_max)
What the solve loop of the solve in flight last saw (`to_solve_progress`), and a blocking wait until at most `active_max` trajectories
are still iterating (`to_solve_wait_below`).  A host with more work than one batch pipelines it over two `BatchProblem`s: the next
solve is admitted when the one in flight has drained (`solve_pipelined!`).
"""
function solve_progress(s::PendingSolve)
    a, b, f = Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0)
    check(ccall((:to_solve_progress, lib), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}), s.p.handle, a, b, f))
    (active = a[], batch_steps = b[], in_flight = f[] != 0)
end
wait_below!(s::PendingSolve, active_max::Integer) =
    check(ccall((:to_solve_wait_below, lib), Cint, (Ptr{Cvoid}, Int32), s.p.handle, Int32(active_max)))
"""
    solve_pipelined!(problems, jobs; which = :ilqr, admit_below = B, prepare! = (p, job) -> nothing)
`jobs` solves over `length(problems)` handles of the same shape: job i runs on `problems[mod1(i, depth)]`, is prepared with
`prepare!(p, i)` (new x0 / goal / initial controls) once that handle's previous solve has been collected, and is admitted as soon as the
job in front of it has at most `admit_below` trajectories still iterating.  Returns the statistics of every job, in order.
"""
function solve_pipelined!(problems::Vector{BatchProblem}, jobs::Integer; which::Symbol = :ilqr,
                          admit_below::Integer = problems[1].B, prepare! = (p, job) -> nothing)
    depth = length(problems)
    pending = Vector{Union{Nothing,PendingSolve}}(nothing, depth)
    out = Vector{Any}(undef, jobs)
    slotjob = zeros(Int, depth)
    last = nothing
    for job in 1:jobs
        slot = mod1(job, depth)
        if pending[slot] !== nothing
            out[slotjob[slot]] = wait_solve!(pending[slot]); pending[slot] = nothing
        end
        prepare!(problems[slot], job)
        last === nothing || wait_below!(last, admit_below)
        last = pending[slot] = solve_async!(problems[slot], which)
        slotjob[slot] = job
    end
    for slot in 1:depth
        pending[slot] === nothing || (out[slotjob[slot]] = wait_solve!(pending[slot]))
    end
    out
end

# ---- expansion / gains (error-state blocks; examples/Internal API.ipynb)
function dynamics_jacobians(p::BatchProblem)
    A, Bm = zeros(p.ne, p.ne, p.N - 1, p.B), zeros(p.ne, p.m, p.N - 1, p.B)
    check(ccall((:to_get_dynamics_jacobians, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.handle, A, Bm))
    (A = A, B = Bm)
end
function cost_expansion(p::BatchProblem)
    Qxx, Quu, Qux = zeros(p.ne, p.ne, p.N, p.B), zeros(p.m, p.m, p.N, p.B), zeros(p.m, p.ne, p.N, p.B)
    qx, qu = zeros(p.ne, p.N, p.B), zeros(p.m, p.N, p.B)
    check(ccall((:to_get_cost_expansion, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        p.handle, Qxx, Quu, Qux, qx, qu))
    (Qxx = Qxx, Quu = Quu, Qux = Qux, qx = qx, qu = qu)
end
function gains(p::BatchProblem)
    K, d, dV, rho = zeros(p.m, p.ne, p.N - 1, p.B), zeros(p.m, p.N - 1, p.B), zeros(2, p.B), zeros(p.B)
    check(ccall((:to_get_gains, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), p.handle, K, d, dV, rho))
    (K = K, d = d, dV = dV, rho = rho)
end
"""
    TO.set_goal_state!(p::BatchProblem, Xf::Matrix)        # Xf :: (n, B): one goal per trajectory
`set_LQR_goal!(cost, xf_b)` (src/cost_functions.jl:249-252: q = -Q xf, nothing else) on every cost of the objective for every
trajectory of the batch — batched MPC / goal sweeps on one handle (SURVEY §8b) — and, with `constraint = true` (the reference's default,
src/problem.jl:303-309), the target `xf_b[inds]` of every GoalConstraint (`to_set_constraint_params_batch`); `clear_goal_state_batch!`
returns to the shared descriptors.
"""
function TO.set_goal_state!(p::BatchProblem, Xf::Matrix{Float64}; objective::Bool = true, constraint::Bool = true)
    size(Xf) == (p.n, p.B) || throw(DimensionMismatch("Xf must be (n, B)"))
    if objective
        costs = Any[]
        for c in p.prob.obj.cost
            any(x -> x === c, costs) || push!(costs, c)
        end
        for (i, c) in enumerate(costs)
            c isa TO.QuadraticCostFunction || throw(MethodError(TO.set_LQR_goal!, (c, Xf)))
            q = -(c.Q * Xf)                                   # (n, B), column-major
            check(ccall((:to_set_cost_linear_batch, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), p.handle, i - 1, q, C_NULL))
        end
    end
    if constraint   # src/problem.jl:303-309: every GoalConstraint of the list gets the new target — here one per trajectory
        for (i, con) in enumerate(p.prob.constraints)
            con isa TO.GoalConstraint || continue
            par = Xf[collect(con.inds), :]                    # (p, B), column-major
            check(ccall((:to_set_constraint_params_batch, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), p.handle, i - 1, par))
        end
    end
    nothing
end
"""
    set_constraint_params_batch!(p, con_id, params)       # params :: (p_rows, B)
One parameter set per trajectory for constraint `con_id` (1-based position in the ConstraintList): a `GoalConstraint`'s target `xf_b[inds]`
or a `LinearConstraint`'s right-hand side `b_b` (`to_set_constraint_params_batch`).
"""
set_constraint_params_batch!(p::BatchProblem, con_id::Integer, params::Matrix{Float64}) =
    check(ccall((:to_set_constraint_params_batch, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), p.handle, Int32(con_id - 1), params))
function clear_goal_state_batch!(p::BatchProblem)
    check(ccall((:to_clear_cost_linear_batch, lib), Cint, (Ptr{Cvoid},), p.handle))
    check(ccall((:to_clear_constraint_params_batch, lib), Cint, (Ptr{Cvoid},), p.handle))
end
"Cost-to-go of the last backward pass: S[ne,ne,N,B], s[ne,N,B] (S_N = Qxx_N; S_k = Qxx + K'Quu K + K'Qux + Qux'K)."
function cost_to_go(p::BatchProblem)
    S, s = zeros(p.ne, p.ne, p.N, p.B), zeros(p.ne, p.N, p.B)
    check(ccall((:to_get_cost_to_go, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.handle, S, s))
    (S = S, s = s)
end
"Altro's infeasible_controls for a handle whose model is an InfeasibleBase: slack controls w_k = x_{k+1} - f_d(x_k, u_k) from the current
states (TO.initial_states!, src/problem.jl:242-253), so that the rollout every solve starts with reproduces the state guess."
infeasible_controls!(p::BatchProblem) = check(ccall((:to_infeasible_controls, lib), Cint, (Ptr{Cvoid},), p.handle))
"RD.gradient! / RD.hessian! of the objective on every knot (src/cost_functions.jl:137-233)."
function cost_gradient_hessian(p::BatchProblem)
    nz = p.n + p.m
    g, H = zeros(nz, p.N, p.B), zeros(nz, nz, p.N, p.B)
    check(ccall((:to_cost_expansion, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), p.handle, g, H))
    (grad = g, hess = H)
end
"RD.jacobian! of the discretised dynamics on every knot: F = [A B]."
function discrete_jacobian(p::BatchProblem)
    F = zeros(p.n, p.n + p.m, p.N - 1, p.B)
    check(ccall((:to_discrete_jacobian, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, F))
    F
end

# ---- constraints (src/abstract_constraint.jl:200-280)
function constraint_info(p::BatchProblem, i::Integer)
    cp, cw, cnk, cs = Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0)
    check(ccall((:to_constraint_info, lib), Cint, (Ptr{Cvoid}, Int32, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}), p.handle, i - 1, cp, cw, cnk, cs))
    (p = Int(cp[]), width = Int(cw[]), nk = Int(cnk[]), sense = Int(cs[]))
end
"evaluate_constraints!: values of constraint `i` (1-based, ConstraintList order) over its knot range, (p, nk, B)."
function TO.evaluate_constraints!(p::BatchProblem, i::Integer)
    c = constraint_info(p, i)
    vals = zeros(c.p, c.nk, p.B)
    check(ccall((:to_evaluate_constraints, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), p.handle, i - 1, vals))
    vals
end
"constraint_jacobians!: (p, w, nk, B) with w = n for state constraints, n+m otherwise; fully written."
function TO.constraint_jacobians!(p::BatchProblem, i::Integer)
    c = constraint_info(p, i)
    jac = zeros(c.p, c.width, c.nk, p.B)
    check(ccall((:to_constraint_jacobians, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), p.handle, i - 1, jac))
    jac
end
"∇constraint_jacobians!: H += Σ_r λ_r ∇²c_r (ADDS, like the reference; src/abstract_constraint.jl:255-280)."
function TO.∇constraint_jacobians!(p::BatchProblem, i::Integer, H::Array{Float64,4}, λ::Array{Float64,3})
    check(ccall((:to_constraint_hessians, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), p.handle, i - 1, λ, H))
    H
end
"max |x_1 ⊖ x0|, |x_{k+1} ⊖ f(x_k, u_k)| per trajectory: 0 for a rollout, the dynamics infeasibility a polish leaves otherwise."
function dynamics_defect(p::BatchProblem)
    d = zeros(p.B)
    check(ccall((:to_dynamics_defect, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, d))
    d
end
function TO.max_violation(p::BatchProblem)
    c = zeros(p.B)
    check(ccall((:to_max_violation, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), p.handle, c))
    c
end
function duals(p::BatchProblem, i::Integer)
    c = constraint_info(p, i)
    λ, μ = zeros(c.p, c.nk, p.B), zeros(p.B)
    check(ccall((:to_get_duals, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), p.handle, i - 1, λ, μ))
    (λ = λ, μ = μ)
end
set_duals!(p::BatchProblem, i::Integer, λ::Array{Float64,3}, μ::Vector{Float64}) =
    check(ccall((:to_set_duals, lib), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}), p.handle, i - 1, λ, μ))
reset_duals!(p::BatchProblem) = check(ccall((:to_reset_duals, lib), Cint, (Ptr{Cvoid},), p.handle))
dual_update!(p::BatchProblem) = check(ccall((:to_dual_update, lib), Cint, (Ptr{Cvoid},), p.handle))

# ---- cones, batched and stateless (src/cones.jl:96-291); x is (dim, count)
function TO.projection!(cone::TO.Conic, px::Matrix{Float64}, x::Matrix{Float64}; device = 0)
    status = zeros(Int32, size(x, 2))
    check(ccall((:to_cone_projection, lib), Cint, (Cint, Int32, Int32, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
        device, conecode(cone), size(x, 1), size(x, 2), x, px, status))
    px
end
function TO.∇projection!(cone::TO.Conic, J::Array{Float64,3}, x::Matrix{Float64}; device = 0)
    check(ccall((:to_cone_projection_jacobian, lib), Cint, (Cint, Int32, Int32, Int64, Ptr{Float64}, Ptr{Float64}),
        device, conecode(cone), size(x, 1), size(x, 2), x, J))
    J
end
function TO.∇²projection!(cone::TO.Conic, H::Array{Float64,3}, x::Matrix{Float64}, b::Matrix{Float64}; device = 0)
    check(ccall((:to_cone_projection_hessian, lib), Cint, (Cint, Int32, Int32, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        device, conecode(cone), size(x, 1), size(x, 2), x, b, H))
    H
end

# ---- multi-GPU: one process per GPU, the batch shards as independent units, one RCCL all-gather of the results
"128-byte RCCL id: create on rank 0, ship to the other ranks (MPI.Bcast!, a file, ...)."
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:to_comm_unique_id, lib), Cint, (Ptr{Cvoid},), id))
    id
end
comm_init_rank!(p::BatchProblem, nranks::Integer, rank::Integer, id::Vector{UInt8}) =
    check(ccall((:to_comm_init_rank, lib), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}), p.handle, nranks, rank, id))
"All ranks' trajectories into caller-owned device buffers of n*N*B_total and m*(N-1)*B_total doubles (global order; shards may differ in size)."
allgather!(p::BatchProblem, dX_all::Ptr{Cvoid}, dU_all::Ptr{Cvoid}) =
    check(ccall((:to_allgather, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), p.handle, dX_all, dU_all))
"(nranks, rank, B_total, counts): the shard sizes exchanged at comm_init_rank!."
function comm_shards(p::BatchProblem)
    nr, rk, tot = Ref{Int32}(0), Ref{Int32}(0), Ref{Int64}(0)
    check(ccall((:to_comm_shards, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}), p.handle, nr, rk, tot, C_NULL))
    counts = zeros(Int32, nr[])
    check(ccall((:to_comm_shards, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}), p.handle, nr, rk, tot, counts))
    (nranks = Int(nr[]), rank = Int(rk[]), B_total = Int(tot[]), counts = counts)
end
"iterations, status and objective cost of every rank's trajectories (host vectors of B_total entries, global order)."
function allgather_stats(p::BatchProblem)
    T = comm_shards(p).B_total
    its, st, J = zeros(Int32, T), zeros(Int32, T), zeros(T)
    check(ccall((:to_allgather_stats, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}), p.handle, its, st, J))
    (iterations = its, status = SolverStatus.(st), cost = J)
end
comm_destroy!(p::BatchProblem) = check(ccall((:to_comm_destroy, lib), Cint, (Ptr{Cvoid},), p.handle))

"(backward = :coop / :mfma / :lane, fused_expansion, compaction, first_round, forward_waves, scan_backward, accept_by_rollout, line_search_repack): the kernels a solve on this handle runs."
function solver_path(p::BatchProblem)
    info = zeros(Int32, 8)
    check(ccall((:to_solver_path, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}), p.handle, info))
    (backward = (:coop, :mfma, :lane)[info[1] + 1], fused_expansion = info[2] != 0, compaction = info[3] != 0, first_round = Int(info[4]),
     forward_waves = Int(info[5]), scan_backward = info[6] != 0, accept_by_rollout = info[7] != 0, line_search_repack = (info[8] & 1) != 0, repacked_working_set = (info[8] & 2) != 0)
end

"""
    knot_dims(p) -> (nx, nu)

Live state / control dimension at each of the N knots (`RD.dims(models)`, src/dynamics.jl:15-31).  `(n, m)` everywhere unless the
handle was created for a hybrid model vector (`TO_MODEL_HYBRID_DOUBLE_INTEGRATOR`, whose states / controls are stored zero-padded at
the largest dimensions).  A Julia host builds such a handle from the descriptor directly: model id 3, `model_params = [mass, S]`
(S = time steps of the first model), costs / constraints of the narrower knots given at the storage dimensions with nothing on
the padding (the Python package's `pad_cost` / `IndexedConstraint` lowering, trajectoryoptimization.jl_amd/api.py, is the recipe).
"""
function knot_dims(p::BatchProblem)
    nx, nu = zeros(Int32, p.N), zeros(Int32, p.N)
    check(ccall((:to_knot_dims, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}), p.handle, nx, nu))
    (Int.(nx), Int.(nu))
end

# ---- measurement
set_profiling!(p::BatchProblem, on::Bool) = check(ccall((:to_set_profiling, lib), Cint, (Ptr{Cvoid}, Cint), p.handle, on))
reset_profile!(p::BatchProblem) = check(ccall((:to_reset_profile, lib), Cint, (Ptr{Cvoid},), p.handle))
function profile(p::BatchProblem)
    ms, launches = zeros(PROFILE_SLOTS), zeros(Int64, PROFILE_SLOTS)
    check(ccall((:to_get_profile, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), p.handle, ms, launches))
    (kernel_ms = ms, launches = launches)
end

export LinearMap, MassDoubleIntegrator, BatchProblem, SolverOpts, solver_options, default_options, solve_ilqr!, solve_al!, solve_pn!, solve_altro!, solve_async!, wait_solve!, dynamics_defect, expand!, backwardpass!, forwardpass!,
    stage_costs, al_cost, dynamics_jacobians, cost_expansion, gains, cost_gradient_hessian, discrete_jacobian, duals, set_duals!,
    reset_duals!, dual_update!, comm_unique_id, comm_init_rank!, allgather!, allgather_stats, comm_shards, comm_destroy!, solver_path, knot_dims, device_count, build_id

end # module
