# RigidBodyDynamicsGPU.jl — thin `ccall` shim that places librbd_hip.so (include/rbd_hip.h) under
# RigidBodyDynamics.jl's own generics for the batched hot path.
#
# NOT EXECUTED IN THIS REPO: the build image has no Julia toolchain (SURVEY.md F3). The file shows the binding a
# maintainer of the reference would add; the tested boundary is the C ABI itself (tests/ drive it through ctypes).
#
# Usage sketch:
#     using RigidBodyDynamics, RigidBodyDynamicsGPU
#     mechanism = parse_urdf("atlas.urdf", floating = true)
#     state  = BatchedMechanismState(mechanism, 4096)            # q :: nq × B, v :: nv × B   (one state per column)
#     result = BatchedDynamicsResult(mechanism, 4096)
#     rand!(state)
#     dynamics!(result, state, τ)                                 # τ :: nv × B ; fills result.v̇, result.q̇
#     inverse_dynamics!(τout, state, v̇) ; mass_matrix!(result, state) ; dynamics_bias!(result, state)
module RigidBodyDynamicsGPU

using RigidBodyDynamics
using RigidBodyDynamics: Mechanism, MechanismState, DynamicsResult, Joint, JointType, Revolute, Prismatic, Fixed,
    QuaternionFloating, SinCosRevolute, tree_joints, non_tree_joints, predecessor, successor, joint_to_predecessor,
    joint_to_successor, spatial_inertia, joint_type, num_positions, num_velocities, num_constraints, root_body, modcount
using RigidBodyDynamics.Spatial: rotation, translation
import RigidBodyDynamics: dynamics!, inverse_dynamics!, mass_matrix!, dynamics_bias!
using LinearAlgebra

export BatchedMechanismState, BatchedDynamicsResult, librbd_hip

const librbd_hip = Ref("librbd_hip.so")   # set to <repo>/rigidbodydynamics.jl_amd/csrc/librbd_hip.so

# ---- status codes -> Julia exceptions (same types the reference throws) --------------------------------------
function check(status::Cint, where::String)
    status == 0 && return nothing
    msg = unsafe_string(ccall((:rbd_status_string, librbd_hip[]), Cstring, (Cint,), status))
    status == 1 && throw(ArgumentError("$where: $msg"))
    status == 2 && throw(DimensionMismatch("$where: $msg"))
    status == 7 && error("This method can currently only handle tree Mechanisms.")   # mechanism_algorithms.jl:549
    status == 8 && throw(PosDefException(0))                                          # LAPACK.potrf!
    error("$where: $msg ($(unsafe_string(ccall((:rbd_last_hip_error, librbd_hip[]), Cstring, ()))))")
end

# ---- C structs (field order == include/rbd_hip.h) -------------------------------------------------------------
struct RbdLoopJoint
    predecessor::Int32; successor::Int32; joint_type::Int32; _pad::Int32
    axis::NTuple{3, Float64}
    pred_rot::NTuple{9, Float64}; pred_trans::NTuple{3, Float64}
    succ_rot::NTuple{9, Float64}; succ_trans::NTuple{3, Float64}
    rotation_from_z_aligned::NTuple{9, Float64}
    gains::NTuple{4, Float64}
end

struct RbdFlatModel
    n_bodies::Int32; nq::Int32; nv::Int32; n_loops::Int32
    parent::Ptr{Int32}; joint_type::Ptr{Int32}; q_offset::Ptr{Int32}; v_offset::Ptr{Int32}
    joint_axis::Ptr{Float64}; joint_axis2::Ptr{Float64}
    pred_rot::Ptr{Float64}; pred_trans::Ptr{Float64}
    inertia_moment::Ptr{Float64}; inertia_cross::Ptr{Float64}; inertia_mass::Ptr{Float64}
    gravity::NTuple{3, Float64}
    loops::Ptr{RbdLoopJoint}
end

struct RbdOpts
    layout::Int32; memory::Int32; algorithm::Int32; stabilization::Int32
end
const LAYOUT_AOS = Int32(1)    # Julia n × B column-major == one state per column
const MEM_DEVICE, MEM_HOST = Int32(0), Int32(1)

jointtag(::Fixed) = Int32(0); jointtag(::Revolute) = Int32(1); jointtag(::Prismatic) = Int32(2)
jointtag(::QuaternionFloating) = Int32(3); jointtag(::Planar) = Int32(4); jointtag(::QuaternionSpherical) = Int32(5)
jointtag(::SinCosRevolute) = Int32(6)
jointtag(jt::JointType) = throw(ArgumentError("joint type $(typeof(jt)) is not supported by librbd_hip (e.g. SPQuatFloating)"))
jointaxis(jt::Union{Revolute, Prismatic, SinCosRevolute}) = Tuple(Float64.(jt.axis))
jointaxis(jt::Planar) = Tuple(Float64.(jt.x_axis))          # planar.jl: translation along x_axis / y_axis, rotation about x × y
jointaxis(::JointType) = (0.0, 0.0, 0.0)
jointaxis2(jt::Planar) = Tuple(Float64.(jt.y_axis))
jointaxis2(::JointType) = (0.0, 0.0, 0.0)
rowmajor(R) = Float64[R[i, j] for i in 1:3 for j in 1:3]    # the C side stores rotations row-major

# ---- flatten once on the host: exactly the tables MechanismState tabulates (mechanism_state.jl:85-118) -----------
mutable struct FlatModelHandle
    handle::Ptr{Cvoid}
    modcount::Int
    nq::Int; nv::Int; nc::Int; nb::Int
    bodyindex::Dict{RigidBody, Int32}   # body -> index in the flat model (-1 = root body)
end

function FlatModelHandle(mechanism::Mechanism)
    tj = collect(tree_joints(mechanism))
    nb = length(tj)
    bodyindex = Dict(successor(j, mechanism) => Int32(i - 1) for (i, j) in enumerate(tj))
    bodyindex[root_body(mechanism)] = Int32(-1)
    parent = Int32[bodyindex[predecessor(j, mechanism)] for j in tj]
    jtype = Int32[jointtag(joint_type(j)) for j in tj]
    qoff = Int32[0; cumsum(num_positions.(tj))[1:end-1]]
    voff = Int32[0; cumsum(num_velocities.(tj))[1:end-1]]
    axis = reduce(vcat, [collect(jointaxis(joint_type(j))) for j in tj]; init = Float64[])
    axis2 = reduce(vcat, [collect(jointaxis2(joint_type(j))) for j in tj]; init = Float64[])
    prot = reduce(vcat, [rowmajor(rotation(joint_to_predecessor(j))) for j in tj]; init = Float64[])
    ptrans = reduce(vcat, [Float64.(translation(joint_to_predecessor(j))) for j in tj]; init = Float64[])
    inertias = [spatial_inertia(successor(j, mechanism)) for j in tj]      # expressed in frame_after(joint) after canonicalization
    moment = reduce(vcat, [rowmajor(I.moment) for I in inertias]; init = Float64[])
    cross = reduce(vcat, [Float64.(I.cross_part) for I in inertias]; init = Float64[])
    mass = Float64[I.mass for I in inertias]
    loops = RbdLoopJoint[]
    for j in non_tree_joints(mechanism)
        jt = joint_type(j)
        Rz = jt isa Union{Revolute, Prismatic} ? rowmajor(jt.rotation_from_z_aligned) :
             jt isa Planar ? rowmajor(hcat(jt.x_axis, jt.y_axis, jt.rot_axis)) :      # columns (x, y, x × y): include/rbd_hip.h
             rowmajor(Matrix(1.0I, 3, 3))
        push!(loops, RbdLoopJoint(bodyindex[predecessor(j, mechanism)], bodyindex[successor(j, mechanism)], jointtag(jt), 0,
            jointaxis(jt), Tuple(rowmajor(rotation(joint_to_predecessor(j)))), Tuple(Float64.(translation(joint_to_predecessor(j)))),
            Tuple(rowmajor(rotation(joint_to_successor(j)))), Tuple(Float64.(translation(joint_to_successor(j)))),
            Tuple(Rz), (100.0, 20.0, 100.0, 20.0)))     # default_constraint_stabilization_gains (mechanism_algorithms.jl:610-612)
    end
    g = mechanism.gravitational_acceleration.v
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve parent jtype qoff voff axis axis2 prot ptrans moment cross mass loops begin
        desc = Ref(RbdFlatModel(nb, sum(num_positions, tj; init = 0), sum(num_velocities, tj; init = 0), length(loops),
            pointer(parent), pointer(jtype), pointer(qoff), pointer(voff), pointer(axis), pointer(axis2), pointer(prot), pointer(ptrans),
            pointer(moment), pointer(cross), pointer(mass), (g[1], g[2], g[3]), isempty(loops) ? C_NULL : pointer(loops)))
        check(ccall((:rbd_model_create, librbd_hip[]), Cint, (Ref{RbdFlatModel}, Ref{Ptr{Cvoid}}), desc, handle), "rbd_model_create")
    end
    m = FlatModelHandle(handle[], modcount(mechanism), num_positions(mechanism), num_velocities(mechanism),
        sum(num_constraints, non_tree_joints(mechanism); init = 0), nb, Dict{RigidBody, Int32}(bodyindex))
    finalizer(x -> ccall((:rbd_model_destroy, librbd_hip[]), Cint, (Ptr{Cvoid},), x.handle), m)
    m
end

# ---- batched state / result: same field names as the reference types -------------------------------------------
mutable struct BatchedMechanismState{T}
    mechanism::Mechanism
    model::FlatModelHandle
    ws::Ptr{Cvoid}
    q::Matrix{T}      # nq × B   (host buffers here; with AMDGPU.jl pass ROCArray pointers and MEM_DEVICE instead)
    v::Matrix{T}      # nv × B
end

function BatchedMechanismState(mechanism::Mechanism, B::Integer; T::Type = Float64, device::Integer = 0)
    model = FlatModelHandle(mechanism)
    ws = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rbd_workspace_create, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
        model.handle, B, device, T === Float64 ? 0 : 1, C_NULL, ws), "rbd_workspace_create")
    s = BatchedMechanismState{T}(mechanism, model, ws[], zeros(T, model.nq, B), zeros(T, model.nv, B))
    finalizer(x -> ccall((:rbd_workspace_destroy, librbd_hip[]), Cint, (Ptr{Cvoid},), x.ws), s)
    s
end

# @modcountcheck (src/util.jl:61): re-flatten when the mechanism was modified
function checkmodcount(state::BatchedMechanismState)
    modcount(state.mechanism) == state.model.modcount || throw(RigidBodyDynamics.ModificationCountMismatch("state out of date with mechanism"))
end

struct BatchedDynamicsResult{T}
    massmatrix::Array{T, 3}     # nv × nv × B, lower triangles valid (Symmetric(…, :L) per state)
    dynamicsbias::Matrix{T}     # nv × B
    q̇::Matrix{T}; v̇::Matrix{T}; λ::Matrix{T}
    constraintjacobian::Array{T, 3}; constraintbias::Matrix{T}
end
function BatchedDynamicsResult(mechanism::Mechanism, B::Integer; T::Type = Float64)
    nq, nv = num_positions(mechanism), num_velocities(mechanism)
    nc = sum(num_constraints, non_tree_joints(mechanism); init = 0)
    BatchedDynamicsResult{T}(zeros(T, nv, nv, B), zeros(T, nv, B), zeros(T, nq, B), zeros(T, nv, B), zeros(T, nc, B), zeros(T, nc, nv, B), zeros(T, nc, B))
end

opts(; algorithm = 0, stabilization = 1, memory = MEM_HOST) = Ref(RbdOpts(LAYOUT_AOS, memory, algorithm, stabilization))
nullable(x::AbstractArray) = pointer(x)
nullable(::Nothing) = C_NULL
batchsize(state) = size(state.q, 2)

# ---- the four generics -----------------------------------------------------------------------------------------
"""`dynamics!(result, state, torques, externalwrenches; stabilization_gains)` — src/mechanism_algorithms.jl:845-864.
`torques`: nv × B or `nothing` (zeros); `externalwrenches`: 6·n_bodies × B root-frame wrenches (torque; force) or `nothing`."""
function dynamics!(result::BatchedDynamicsResult{T}, state::BatchedMechanismState{T}, torques = nothing, externalwrenches = nothing;
        stabilization_gains = :default, algorithm::Symbol = :aba) where {T}
    checkmodcount(state)
    B = batchsize(state)
    torques === nothing || size(torques) == (state.model.nv, B) || throw(DimensionMismatch("torques"))
    o = opts(algorithm = algorithm === :aba ? 0 : 1, stabilization = stabilization_gains === nothing ? 0 : 1)
    λptr = state.model.nc > 0 ? pointer(result.λ) : C_NULL
    check(ccall((:rbd_dynamics, librbd_hip[]), Cint,
        (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, nullable(torques), nullable(externalwrenches), result.v̇, result.q̇, λptr, o), "rbd_dynamics")
    if algorithm !== :aba || state.model.nc > 0
        check(ccall((:rbd_dynamics_result, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
            state.ws, B, result.massmatrix, result.dynamicsbias, state.model.nc > 0 ? pointer(result.constraintjacobian) : C_NULL,
            state.model.nc > 0 ? pointer(result.constraintbias) : C_NULL, o), "rbd_dynamics_result")
    end
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    nothing
end

"""`inverse_dynamics!(torquesout, jointwrenchesout, accelerations, state, v̇, externalwrenches)` — :542-553 (the per-body
wrench/acceleration dictionaries are workspace-internal on the device)."""
function inverse_dynamics!(torquesout::Matrix{T}, state::BatchedMechanismState{T}, v̇::Matrix{T}, externalwrenches = nothing) where {T}
    checkmodcount(state)
    B = batchsize(state)
    size(torquesout) == (state.model.nv, B) || error("length of torque vector is wrong")
    check(ccall((:rbd_inverse_dynamics, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, v̇, nullable(externalwrenches), torquesout, opts()), "rbd_inverse_dynamics")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    torquesout
end

"""`dynamics_bias!(result, state)` — :496-498."""
function dynamics_bias!(result::BatchedDynamicsResult{T}, state::BatchedMechanismState{T}, externalwrenches = nothing) where {T}
    checkmodcount(state)
    check(ccall((:rbd_dynamics_bias, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, batchsize(state), state.q, state.v, nullable(externalwrenches), result.dynamicsbias, opts()), "rbd_dynamics_bias")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    result.dynamicsbias
end

"""`mass_matrix!(M, state)` / `mass_matrix!(result, state)` — :248-274; lower triangles written (uplo == 'L')."""
function mass_matrix!(M::Array{T, 3}, state::BatchedMechanismState{T}) where {T}
    checkmodcount(state)
    nv, B = state.model.nv, batchsize(state)
    size(M) == (nv, nv, B) || throw(DimensionMismatch("mass matrix has wrong size"))
    check(ccall((:rbd_mass_matrix, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, M, opts()), "rbd_mass_matrix")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    M
end
mass_matrix!(result::BatchedDynamicsResult, state::BatchedMechanismState) = mass_matrix!(result.massmatrix, state)

"""`simulate(state0, final_time; Δt, stabilization_gains)` — src/simulate.jl:36-55 for the whole batch: Munthe-Kaas RK4 on the
device (`rbd_simulate`), constant `torques` (the default control is `zero_torque!`); `state.q`, `state.v` are advanced in place."""
function RigidBodyDynamics.simulate(state::BatchedMechanismState{T}, final_time; Δt = 1e-4, torques = nothing, stabilization_gains = :default) where {T}
    checkmodcount(state)
    nsteps, t = 0, zero(T)
    while t < final_time            # same loop as integrate(), src/ode_integrators.jl:311-314
        t += Δt; nsteps += 1
    end
    check(ccall((:rbd_simulate, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Cdouble, Int32, Ref{RbdOpts}),
        state.ws, batchsize(state), state.q, state.v, nullable(torques), C_NULL, Float64(Δt), nsteps,
        opts(stabilization = stabilization_gains === nothing ? 0 : 1)), "rbd_simulate")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    range(zero(T), step = T(Δt), length = nsteps + 1)
end


# ---- kinematics by-products of the same forward-kinematics pass (root frame; host buffers like the methods above) ---------------
# momentum_matrix!(out, state) mechanism_algorithms.jl:313-327: out is 6 × nv × B
function momentum_matrix!(out::Array{T, 3}, state::BatchedMechanismState{T}) where {T}
    checkmodcount(state)
    B = size(state.q, 2)
    size(out) == (6, state.model.nv, B) || throw(DimensionMismatch())
    check(ccall((:rbd_kinematics, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, out, C_NULL, C_NULL, opts()), "rbd_kinematics")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    out
end

# center_of_mass(state) :28-50 -> 3 × B;  kinetic_energy / gravitational_potential_energy mechanism_state.jl:886-903 -> B each
function center_of_mass(state::BatchedMechanismState{T}) where {T}
    B = size(state.q, 2); com = Matrix{T}(undef, 3, B)
    check(ccall((:rbd_kinematics, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, C_NULL, com, C_NULL, opts()), "rbd_kinematics")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    com
end
function energies(state::BatchedMechanismState{T}) where {T}   # row 1: kinetic_energy, row 2: gravitational_potential_energy
    B = size(state.q, 2); e = Matrix{T}(undef, 2, B)
    check(ccall((:rbd_kinematics, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, C_NULL, C_NULL, e, opts()), "rbd_kinematics")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    e
end
kinetic_energy(state::BatchedMechanismState) = energies(state)[1, :]
gravitational_potential_energy(state::BatchedMechanismState) = energies(state)[2, :]

# momentum(state), momentum_rate_bias(state) mechanism_state.jl:975-987 -> 6 × B each
function momenta(state::BatchedMechanismState{T}) where {T}
    B = size(state.q, 2); out = Matrix{T}(undef, 12, B)
    check(ccall((:rbd_momentum, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}), state.ws, B, state.q, state.v, out, opts()),
        "rbd_momentum")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    out
end
momentum(state::BatchedMechanismState) = momenta(state)[1:6, :]
momentum_rate_bias(state::BatchedMechanismState) = momenta(state)[7:12, :]

# geometric_jacobian!(out, state, path) :80-99 in the root frame; the path is given by its end bodies (path(mechanism, base, body))
function geometric_jacobian!(out::Array{T, 3}, state::BatchedMechanismState{T}, base::RigidBody, body::RigidBody) where {T}
    checkmodcount(state)
    B = size(state.q, 2)
    size(out) == (6, state.model.nv, B) || throw(DimensionMismatch())
    check(ccall((:rbd_geometric_jacobian, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Int32, Int32, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.model.bodyindex[base], state.model.bodyindex[body], out, opts()), "rbd_geometric_jacobian")
    check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
    out
end

end # module
