# RigidBodyDynamicsGPU.jl — thin `ccall` shim that places librbd_hip.so (include/rbd_hip.h) under
# RigidBodyDynamics.jl's own generics for the batched hot path.
#
# EXPERIMENTAL / NOT EXECUTED IN THIS REPO: the build image has no Julia toolchain (SURVEY.md F3).  The file shows the binding a
# maintainer of the reference would add; the tested boundary is the C ABI itself (tests/ drive it through ctypes), and
# scripts/check_julia_ccalls.py (run by tests/test_capi_symbols.py) checks every `ccall` here against include/rbd_hip.h — symbol
# name, argument count, pointer-vs-integer kind of every argument — so the shim cannot drift from the header unnoticed.
#
# Usage sketch:
#     using RigidBodyDynamics, RigidBodyDynamicsGPU
#     mechanism = parse_urdf("atlas.urdf", floating = true)
#     state  = BatchedMechanismState(mechanism, 4096)            # q :: nq × B, v :: nv × B (one state per column), DEVICE resident
#     result = BatchedDynamicsResult(mechanism, 4096)
#     copyto!(state.q, qhost); copyto!(state.v, vhost)           # or storage = :host for plain Matrix buffers (PCIe inside every call)
#     dynamics!(result, state, τ)                                 # τ :: nv × B ; fills result.v̇, result.q̇ (asynchronous; synchronize(state))
#     inverse_dynamics!(τout, jointwrenches, accelerations, state, v̇, externalwrenches)
#     mass_matrix!(result, state) ; dynamics_bias!(result, state) ; dynamics!(ẋ, result, state, x)
module RigidBodyDynamicsGPU

using RigidBodyDynamics
using RigidBodyDynamics: Mechanism, MechanismState, DynamicsResult, Joint, JointType, Revolute, Prismatic, Fixed, Planar, QuaternionSpherical,
    QuaternionFloating, SinCosRevolute, RigidBody, BodyID, Wrench, tree_joints, non_tree_joints, predecessor, successor, joint_to_predecessor,
    joint_to_successor, spatial_inertia, joint_type, num_positions, num_velocities, num_constraints, root_body, modcount, bodies, JointID
using RigidBodyDynamics.PDControl: SE3PDGains
using RigidBodyDynamics.Spatial: rotation, translation, angular, linear
# the generics this file adds methods to (src/RigidBodyDynamics.jl:143-155 exports them)
import RigidBodyDynamics: dynamics!, inverse_dynamics!, mass_matrix!, dynamics_bias!, momentum_matrix!, geometric_jacobian!, center_of_mass,
    kinetic_energy, gravitational_potential_energy, momentum, momentum_rate_bias, simulate
using LinearAlgebra

export BatchedMechanismState, BatchedDynamicsResult, DeviceMatrix, RbdComm, gather!, gatherv!, mass_matrix_solve_packed!, synchronize, librbd_hip, TorqueTable, PDControl

const librbd_hip = Ref("librbd_hip.so")   # set to <repo>/rigidbodydynamics.jl_amd/csrc/librbd_hip.so
const libhip = Ref("libamdhip64.so")

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

# ---- device-resident n × B buffers without a GPU array package: hipMalloc / hipMemcpy straight from libamdhip64 -------------------
mutable struct DeviceMatrix{T}
    ptr::Ptr{T}
    dims::NTuple{2, Int}
end
function DeviceMatrix{T}(n::Integer, B::Integer) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:hipMalloc, libhip[]), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, max(1, n * B) * sizeof(T)) == 0 || error("hipMalloc failed")
    ccall((:hipMemset, libhip[]), Cint, (Ptr{Cvoid}, Cint, Csize_t), p[], 0, max(1, n * B) * sizeof(T))
    d = DeviceMatrix{T}(Ptr{T}(p[]), (Int(n), Int(B)))
    finalizer(x -> ccall((:hipFree, libhip[]), Cint, (Ptr{Cvoid},), x.ptr), d)
    d
end
Base.size(d::DeviceMatrix) = d.dims
Base.size(d::DeviceMatrix, i) = d.dims[i]
Base.pointer(d::DeviceMatrix) = d.ptr
Base.unsafe_convert(::Type{Ptr{T}}, d::DeviceMatrix{T}) where {T} = d.ptr
function Base.copyto!(d::DeviceMatrix{T}, h::AbstractMatrix{T}) where {T}     # host -> device
    size(h) == d.dims || throw(DimensionMismatch())
    hc = Matrix{T}(h)
    ccall((:hipMemcpy, libhip[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), d.ptr, hc, sizeof(hc), 1) == 0 || error("hipMemcpy failed")
    d
end
function Base.copyto!(h::Matrix{T}, d::DeviceMatrix{T}) where {T}             # device -> host
    size(h) == d.dims || throw(DimensionMismatch())
    ccall((:hipMemcpy, libhip[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), h, d.ptr, sizeof(h), 2) == 0 || error("hipMemcpy failed")
    h
end
Base.Array(d::DeviceMatrix{T}) where {T} = copyto!(Matrix{T}(undef, d.dims...), d)
const Buffer{T} = Union{Matrix{T}, DeviceMatrix{T}}
newbuffer(::Val{:device}, ::Type{T}, n, B) where {T} = DeviceMatrix{T}(n, B)
newbuffer(::Val{:host}, ::Type{T}, n, B) where {T} = zeros(T, n, B)

# ---- C structs (field order == include/rbd_hip.h) -------------------------------------------------------------
struct RbdLoopJoint
    predecessor::Int32; successor::Int32; joint_type::Int32; _pad::Int32
    axis::NTuple{3, Float64}
    pred_rot::NTuple{9, Float64}; pred_trans::NTuple{3, Float64}
    succ_rot::NTuple{9, Float64}; succ_trans::NTuple{3, Float64}
    rotation_from_z_aligned::NTuple{9, Float64}
    gains::NTuple{4, Float64}
end

struct RbdContactPoint        # DefaultContactPoint: src/contact.jl:72-77, :196
    body::Int32; _pad::Int32
    location::NTuple{3, Float64}
    hc_k::Float64; hc_lambda::Float64; hc_n::Float64     # HuntCrossleyModel
    mu::Float64; k::Float64; b::Float64                  # ViscoelasticCoulombModel
end
struct RbdHalfSpace           # HalfSpace3D: src/contact.jl:202-222
    point::NTuple{3, Float64}; outward_normal::NTuple{3, Float64}
end

struct RbdFlatModel
    n_bodies::Int32; nq::Int32; nv::Int32; n_loops::Int32
    parent::Ptr{Int32}; joint_type::Ptr{Int32}; q_offset::Ptr{Int32}; v_offset::Ptr{Int32}
    joint_axis::Ptr{Float64}; joint_axis2::Ptr{Float64}
    pred_rot::Ptr{Float64}; pred_trans::Ptr{Float64}
    inertia_moment::Ptr{Float64}; inertia_cross::Ptr{Float64}; inertia_mass::Ptr{Float64}
    gravity::NTuple{3, Float64}
    loops::Ptr{RbdLoopJoint}
    n_contact_points::Int32; n_halfspaces::Int32
    contact_points::Ptr{RbdContactPoint}; halfspaces::Ptr{RbdHalfSpace}
end

struct RbdOpts
    layout::Int32; memory::Int32; algorithm::Int32; stabilization::Int32
end
struct RbdControl             # device-side controllers of rbd_simulate_controlled (include/rbd_hip.h)
    kind::Int32; per_stage::Int32
    tau::Ptr{Cvoid}; q_des::Ptr{Cvoid}; kp::Ptr{Cvoid}; kd::Ptr{Cvoid}
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
    bodyids::Dict{BodyID, Int32}
    loopjointids::Vector{JointID}       # the non-tree joints in the order of rbd_flat_model_t.loops (= the order of rbd_workspace_set_loop_gains)
end

const WAIT_HOOK = Ref(false)
const HEADER_VERSION = 600    # RBD_HIP_H_VERSION of the include/rbd_hip.h these structs mirror (rbd_flat_model_t grew its contact fields at 200; 400: rbd_workspace_set_loop_gains; 500: rbd_mass_matrix_solve_packed, rbd_gatherv; 600: rbd_jit_check_walk_object)

function FlatModelHandle(mechanism::Mechanism)
    ccall((:rbd_version, librbd_hip[]), Cint, ()) == HEADER_VERSION ||
        error("librbd_hip.so reports another header version than this shim was written for ($HEADER_VERSION): struct layouts may differ")
    if !WAIT_HOOK[]  # once: a kernel compilation still running in the background is waited for before the process tears the compiler down
        atexit(() -> ccall((:rbd_jit_wait_idle, librbd_hip[]), Cvoid, ()))
        WAIT_HOOK[] = true
    end
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
        # rotation_from_z_aligned: revolute.jl:12-17, prismatic.jl, sin_cos_revolute.jl:13 (all three carry the field; the C side builds
        # the constraint wrench basis of a loop joint from its columns)
        Rz = jt isa Union{Revolute, Prismatic, SinCosRevolute} ? rowmajor(jt.rotation_from_z_aligned) :
             jt isa Planar ? rowmajor(hcat(jt.x_axis, jt.y_axis, jt.rot_axis)) :      # columns (x, y, x × y): include/rbd_hip.h
             rowmajor(Matrix(1.0I, 3, 3))
        push!(loops, RbdLoopJoint(bodyindex[predecessor(j, mechanism)], bodyindex[successor(j, mechanism)], jointtag(jt), 0,
            jointaxis(jt), Tuple(rowmajor(rotation(joint_to_predecessor(j)))), Tuple(Float64.(translation(joint_to_predecessor(j)))),
            Tuple(rowmajor(rotation(joint_to_successor(j)))), Tuple(Float64.(translation(joint_to_successor(j)))),
            Tuple(Rz), (100.0, 20.0, 100.0, 20.0)))     # default_constraint_stabilization_gains (mechanism_algorithms.jl:610-612)
    end
    # soft contact: points in the order of the additional state (for body in bodies(m), for point: mechanism_state.jl:143), half-spaces
    cps = RbdContactPoint[]
    for body in bodies(mechanism), point in RigidBodyDynamics.contact_points(body)
        model = RigidBodyDynamics.Contact.contact_model(point)
        loc = RigidBodyDynamics.Contact.location(point).v
        push!(cps, RbdContactPoint(bodyindex[body], 0, (loc[1], loc[2], loc[3]), model.normal.k, model.normal.λ, model.normal.n,
            model.friction.μ, model.friction.k, model.friction.b))
    end
    hss = [RbdHalfSpace(Tuple(Float64.(h.point.v)), Tuple(Float64.(h.outward_normal.v))) for h in mechanism.environment.halfspaces]
    g = mechanism.gravitational_acceleration.v
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve parent jtype qoff voff axis axis2 prot ptrans moment cross mass loops cps hss begin
        desc = Ref(RbdFlatModel(nb, sum(num_positions, tj; init = 0), sum(num_velocities, tj; init = 0), length(loops),
            pointer(parent), pointer(jtype), pointer(qoff), pointer(voff), pointer(axis), pointer(axis2), pointer(prot), pointer(ptrans),
            pointer(moment), pointer(cross), pointer(mass), (g[1], g[2], g[3]), isempty(loops) ? C_NULL : pointer(loops),
            length(cps), length(hss), isempty(cps) ? C_NULL : pointer(cps), isempty(hss) ? C_NULL : pointer(hss)))
        check(ccall((:rbd_model_create, librbd_hip[]), Cint, (Ref{RbdFlatModel}, Ref{Ptr{Cvoid}}), desc, handle), "rbd_model_create")
    end
    m = FlatModelHandle(handle[], modcount(mechanism), num_positions(mechanism), num_velocities(mechanism),
        sum(num_constraints, non_tree_joints(mechanism); init = 0), nb, Dict{RigidBody, Int32}(bodyindex),
        Dict{BodyID, Int32}(BodyID(b) => i for (b, i) in bodyindex), JointID[JointID(j) for j in non_tree_joints(mechanism)])
    finalizer(x -> ccall((:rbd_model_destroy, librbd_hip[]), Cint, (Ptr{Cvoid},), x.handle), m)
    m
end

# Run-time specialised kernels (csrc/rbd_jit.hip): `mass_matrix!` at large batches runs code compiled for this mechanism by hiprtc the first time a
# workspace needs it; `precompile_kernels` pays for the compilation up front (no device needed, the code object is cached on disk).
function precompile_kernels(mechanism::Mechanism; T::Type = Float64)
    model = FlatModelHandle(mechanism)
    log = Vector{UInt8}(undef, 1 << 16)
    status = ccall((:rbd_jit_precompile, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{UInt8}, Int64), model.handle, T === Float64 ? 0 : 1, log, length(log))
    status == 3 && return false    # RBD_ERR_UNSUPPORTED: outside the specialised kernels' scope, or no hiprtc — the generic kernels run
    status == 0 || error("rbd_jit_precompile: " * unsafe_string(pointer(log)))
    true
end

# ---- batched state / result: same field names as the reference types -------------------------------------------
mutable struct BatchedMechanismState{T, A <: Buffer{T}}
    mechanism::Mechanism
    model::FlatModelHandle
    ws::Ptr{Cvoid}
    memory::Int32     # MEM_DEVICE: q, v (and every buffer handed to a call) are device pointers — the default; MEM_HOST: plain Matrix
    q::A              # nq × B
    v::A              # nv × B
    s::A              # num_additional_states × B: the soft-contact friction states (mechanism_state.jl:64, :139-152)
end

"""`storage = :device` (default): q, v are `DeviceMatrix` — resident in HBM, calls are asynchronous on the workspace's stream;
`storage = :host`: plain `Matrix` buffers, staged through PCIe inside every call (convenient, ~5× slower than the kernel)."""
function BatchedMechanismState(mechanism::Mechanism, B::Integer; T::Type = Float64, device::Integer = 0, storage::Symbol = :device)
    model = FlatModelHandle(mechanism)
    ws = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rbd_workspace_create, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
        model.handle, B, device, T === Float64 ? 0 : 1, C_NULL, ws), "rbd_workspace_create")
    q, v = newbuffer(Val(storage), T, model.nq, B), newbuffer(Val(storage), T, model.nv, B)
    s = BatchedMechanismState{T, typeof(q)}(mechanism, model, ws[], storage === :device ? MEM_DEVICE : MEM_HOST, q, v,
        newbuffer(Val(storage), T, RigidBodyDynamics.num_additional_states(mechanism), B))
    finalizer(x -> ccall((:rbd_workspace_destroy, librbd_hip[]), Cint, (Ptr{Cvoid},), x.ws), s)
    s
end

"wait for the asynchronous calls issued on this state's workspace; PosDefException if a mass matrix was not positive definite"
synchronize(state::BatchedMechanismState) = check(ccall((:rbd_sync, librbd_hip[]), Cint, (Ptr{Cvoid},), state.ws), "rbd_sync")
finish(state::BatchedMechanismState) = state.memory == MEM_HOST ? synchronize(state) : nothing   # host buffers must be complete on return

# @modcountcheck (src/util.jl:61): re-flatten when the mechanism was modified
function checkmodcount(state::BatchedMechanismState)
    modcount(state.mechanism) == state.model.modcount || throw(RigidBodyDynamics.ModificationCountMismatch("state out of date with mechanism"))
end

struct BatchedDynamicsResult{T, A <: Buffer{T}}
    massmatrix::A               # (nv·nv) × B: per state an nv × nv column-major matrix, lower triangle valid (Symmetric(…, :L))
    dynamicsbias::A             # nv × B
    q̇::A; v̇::A; λ::A
    constraintjacobian::A; constraintbias::A
    accelerations::A            # (6·n_bodies) × B — src/dynamics_result.jl:26-29, root frame, (angular; linear) per body
    jointwrenches::A
    totalwrenches::A
    ṡ::A                        # num_additional_states × B (dynamics_result.jl:20)
    contactwrenches::A          # (6·n_bodies) × B (dynamics_result.jl:25)
end
function BatchedDynamicsResult(mechanism::Mechanism, B::Integer; T::Type = Float64, storage::Symbol = :device)
    nq, nv, nb = num_positions(mechanism), num_velocities(mechanism), length(collect(tree_joints(mechanism)))
    nc = sum(num_constraints, non_tree_joints(mechanism); init = 0)
    z(n) = newbuffer(Val(storage), T, n, B)
    r = (z(nv * nv), z(nv), z(nq), z(nv), z(nc), z(nc * nv), z(nc), z(6nb), z(6nb), z(6nb), z(RigidBodyDynamics.num_additional_states(mechanism)), z(6nb))
    BatchedDynamicsResult{T, typeof(r[1])}(r...)
end

opts(state; algorithm = 0, stabilization = 1) = Ref(RbdOpts(LAYOUT_AOS, state.memory, algorithm, stabilization))

# `stabilization_gains` (src/mechanism_algorithms.jl:614-632, :848; src/simulate.jl:37) -> the flag of rbd_opts_t + rbd_workspace_set_loop_gains:
#   nothing                                   no Baumgarte stabilization;
#   :default                                  the model's gains = default_constraint_stabilization_gains (100, 20, 100, 20);
#   AbstractDict{JointID, <:SE3PDGains}       per loop joint (a ConstDict{JointID} is one — its getindex ignores the key); a missing joint throws the
#                                             dictionary's own KeyError, as stabilization_gains[nontreejointid] (:655) does in the reference.
# Scalar gains only: the kernels apply k and d as scalars (every use in the reference and its tests); matrix gains throw an ArgumentError
# instead of being silently replaced.  The call is cheap when the gains are the ones already in force.
scalargain(x::Number) = Float64(x)
scalargain(x) = throw(ArgumentError("stabilization_gains: only scalar PD gains are supported by librbd_hip, got $(typeof(x))"))
function stabilization(state, ::Nothing)
    Int32(0)
end
function stabilization(state, gains::Symbol)
    gains === :default || throw(ArgumentError("stabilization_gains: $gains"))
    isempty(state.model.loopjointids) || check(ccall((:rbd_workspace_set_loop_gains, librbd_hip[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), state.ws, C_NULL),
        "rbd_workspace_set_loop_gains")
    Int32(1)
end
function stabilization(state, gains::AbstractDict{JointID, <:SE3PDGains})
    ids = state.model.loopjointids
    isempty(ids) && return Int32(1)
    g = Float64[]
    for id in ids
        gj = gains[id]
        append!(g, (scalargain(gj.angular.k), scalargain(gj.angular.d), scalargain(gj.linear.k), scalargain(gj.linear.d)))
    end
    check(ccall((:rbd_workspace_set_loop_gains, librbd_hip[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), state.ws, g), "rbd_workspace_set_loop_gains")
    Int32(1)
end
stabilization(state, gains) = throw(ArgumentError("stabilization_gains must be nothing or an AbstractDict{JointID, <:SE3PDGains}, got $(typeof(gains))"))
nullable(x) = pointer(x)
nullable(::Nothing) = C_NULL
batchsize(state) = size(state.q, 2)

# externalwrenches: `nothing` (NullDict), a dense (6·n_bodies) × B buffer of root-frame wrenches (torque; force) per body, or the
# reference's own `AbstractDict{BodyID, <:Wrench}` (root frame; applied to every state of the batch) which is densified here
function densewrenches(state::BatchedMechanismState{T}, w::AbstractDict{BodyID, <:Wrench}) where {T}
    nb, B = state.model.nb, batchsize(state)
    h = zeros(T, 6nb, B)
    for (id, wrench) in w
        i = state.model.bodyids[id]
        i < 0 && continue
        h[6i+1:6i+3, :] .= angular(wrench); h[6i+4:6i+6, :] .= linear(wrench)
    end
    state.memory == MEM_HOST ? h : copyto!(DeviceMatrix{T}(6nb, B), h)
end
densewrenches(state, w) = w

# ---- the four generics -----------------------------------------------------------------------------------------
"""`dynamics!(result, state, torques, externalwrenches; stabilization_gains)` — src/mechanism_algorithms.jl:845-864.
`torques`: nv × B or `nothing` (zeros).  Fills result.v̇, q̇ (λ, M, c, K, k on the reference's CRBA route) and, like the reference, the
per-body fields: totalwrenches, and the bias accelerations / joint wrenches of its dynamics_bias! call."""
function dynamics!(result::BatchedDynamicsResult{T}, state::BatchedMechanismState{T}, torques = nothing, externalwrenches = nothing;
        stabilization_gains = :default, algorithm::Symbol = :aba, bodies::Bool = false) where {T}
    checkmodcount(state)
    B = batchsize(state)
    torques === nothing || size(torques) == (state.model.nv, B) || throw(DimensionMismatch("torques"))
    wext = densewrenches(state, externalwrenches)
    o = opts(state; algorithm = algorithm === :aba ? 0 : 1, stabilization = stabilization(state, stabilization_gains))
    λptr = state.model.nc > 0 ? pointer(result.λ) : C_NULL
    if size(state.s, 1) > 0
        # contact points: contact_dynamics! (:680-723), totalwrenches = externalwrenches + contactwrenches (:851-856), then forward dynamics.
        # state.s is reset where a point is not in contact, exactly as the reference resets the contact state in place.  Device buffers only.
        check(ccall((:rbd_dynamics_contact, librbd_hip[]), Cint,
            (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
            state.ws, B, state.q, state.v, state.s, nullable(torques), nullable(wext), result.v̇, result.q̇, result.ṡ, result.contactwrenches,
            result.totalwrenches, o), "rbd_dynamics_contact")
        finish(state)
        return nothing
    end
    # the reference's route on a tree mechanism with device buffers: M and c are written into the result's own buffers (no copy out of the workspace)
    bind = algorithm !== :aba && state.model.nc == 0 && state.memory == MEM_DEVICE
    bind && check(ccall((:rbd_workspace_bind_result, librbd_hip[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), state.ws, result.massmatrix.ptr, result.dynamicsbias.ptr),
        "rbd_workspace_bind_result")
    try
        check(ccall((:rbd_dynamics, librbd_hip[]), Cint,
            (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
            state.ws, B, state.q, state.v, nullable(torques), nullable(wext), result.v̇, result.q̇, λptr, o), "rbd_dynamics")
    finally     # never leave the result's buffers bound to the workspace: a later call would write into them, freed or not
        bind && ccall((:rbd_workspace_bind_result, librbd_hip[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), state.ws, C_NULL, C_NULL)
    end
    if !bind && (algorithm !== :aba || state.model.nc > 0)
        check(ccall((:rbd_dynamics_result, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
            state.ws, B, result.massmatrix, result.dynamicsbias, state.model.nc > 0 ? pointer(result.constraintjacobian) : C_NULL,
            state.model.nc > 0 ? pointer(result.constraintbias) : C_NULL, o), "rbd_dynamics_result")
    end
    if bodies   # result.accelerations / jointwrenches as dynamics!'s own dynamics_bias! call leaves them (:851-856)
        check(ccall((:rbd_dynamics_bias_bodies, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
            state.ws, B, state.q, state.v, nullable(wext), result.dynamicsbias, result.jointwrenches, result.accelerations, opts(state)), "rbd_dynamics_bias_bodies")
    end
    finish(state)
    nothing
end

"""`dynamics!(ẋ, result, state, x, torques, externalwrenches; stabilization_gains)` — the ODE form, :880-889: x = [q; v] per column,
ẋ = [q̇; v̇] (no additional state: there are no contact points)."""
function dynamics!(ẋ::Matrix{T}, result::BatchedDynamicsResult{T}, state::BatchedMechanismState{T}, x::Matrix{T}, torques = nothing,
        externalwrenches = nothing; stabilization_gains = :default) where {T}
    nq, nv = state.model.nq, state.model.nv
    size(x) == (nq + nv, batchsize(state)) || throw(DimensionMismatch("x"))
    copyto!(state.q, x[1:nq, :]); copyto!(state.v, x[nq+1:end, :])          # copyto!(state, x), mechanism_state.jl:465-491
    dynamics!(result, state, torques, externalwrenches; stabilization_gains = stabilization_gains)
    synchronize(state)
    ẋ[1:nq, :] .= result.q̇ isa Matrix ? result.q̇ : Array(result.q̇)          # copyto!(ẋ, result), dynamics_result.jl:89
    ẋ[nq+1:end, :] .= result.v̇ isa Matrix ? result.v̇ : Array(result.v̇)
    ẋ
end

"""`inverse_dynamics!(torquesout, jointwrenchesout, accelerations, state, v̇, externalwrenches)` — :542-553, the reference's own
arity: `jointwrenchesout` / `accelerations` are (6·n_bodies) × B buffers (root frame) or `nothing`."""
function inverse_dynamics!(torquesout::Buffer{T}, jointwrenchesout, accelerations, state::BatchedMechanismState{T}, v̇::Buffer{T},
        externalwrenches = nothing) where {T}
    checkmodcount(state)
    B = batchsize(state)
    size(torquesout) == (state.model.nv, B) || error("length of torque vector is wrong")
    check(ccall((:rbd_inverse_dynamics_bodies, librbd_hip[]), Cint,
        (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, v̇, nullable(densewrenches(state, externalwrenches)), torquesout, nullable(jointwrenchesout),
        nullable(accelerations), opts(state)), "rbd_inverse_dynamics_bodies")
    finish(state)
    torquesout
end
# the short form many callers use: inverse_dynamics!(τ, state, v̇, wext)
inverse_dynamics!(torquesout::Buffer{T}, state::BatchedMechanismState{T}, v̇::Buffer{T}, externalwrenches = nothing) where {T} =
    inverse_dynamics!(torquesout, nothing, nothing, state, v̇, externalwrenches)

"""`dynamics_bias!(result, state)` — :496-498 (external wrenches = result.totalwrenches, as in the reference)."""
function dynamics_bias!(result::BatchedDynamicsResult{T}, state::BatchedMechanismState{T}, externalwrenches = nothing) where {T}
    checkmodcount(state)
    check(ccall((:rbd_dynamics_bias_bodies, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, batchsize(state), state.q, state.v, nullable(densewrenches(state, externalwrenches)), result.dynamicsbias, result.jointwrenches,
        result.accelerations, opts(state)), "rbd_dynamics_bias_bodies")
    finish(state)
    result.dynamicsbias
end

"""`mass_matrix!(M, state)` / `mass_matrix!(result, state)` — :248-274; lower triangles written (uplo == 'L')."""
function mass_matrix!(M::Buffer{T}, state::BatchedMechanismState{T}) where {T}
    checkmodcount(state)
    nv, B = state.model.nv, batchsize(state)
    size(M) == (nv * nv, B) || throw(DimensionMismatch("mass matrix has wrong size"))
    check(ccall((:rbd_mass_matrix, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, M, opts(state)), "rbd_mass_matrix")
    finish(state)
    M
end
mass_matrix!(result::BatchedDynamicsResult, state::BatchedMechanismState) = mass_matrix!(result.massmatrix, state)

"""`mass_matrix_solve_packed!(x, Mpacked, state, rhs)` — `mass_matrix!` + the potrf!/potrs! of `dynamics_solve!` (:764, :819) with M as LAPACK's packed lower
triangle: `Mpacked` is nv(nv+1)/2 × B, element (i, j), i ≥ j (0-based), of a state at i + j(2nv − j − 1)/2 — what `Symmetric(M, :L)` defines, half the bytes of
the square (`rbd_mass_matrix_solve_packed`)."""
function mass_matrix_solve_packed!(x::Buffer{T}, Mpacked::Buffer{T}, state::BatchedMechanismState{T}, rhs::Buffer{T}) where {T}
    checkmodcount(state)
    nv, B = state.model.nv, batchsize(state)
    size(Mpacked) == (nv * (nv + 1) ÷ 2, B) && size(x) == (nv, B) && size(rhs) == (nv, B) || throw(DimensionMismatch("packed mass matrix / x / rhs have wrong sizes"))
    check(ccall((:rbd_mass_matrix_solve_packed, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, rhs, x, Mpacked, opts(state; algorithm = 1)), "rbd_mass_matrix_solve_packed")   # 1 = RBD_ALGO_CRBA_CHOLESKY
    finish(state)
    x
end

# number of steps of integrate(), src/ode_integrators.jl:311-314: `while t < final_time` in the state's scalar type
function stepcount(::Type{T}, final_time, Δt) where {T}
    nsteps, t = 0, zero(T)
    while t < final_time
        t += Δt; nsteps += 1
    end
    nsteps
end
hostcopy(x::Matrix) = copy(x)
hostcopy(x::DeviceMatrix) = Array(x)

"""`simulate(state0, final_time; Δt, stabilization_gains)` — src/simulate.jl:36-55 for the whole batch: Munthe-Kaas RK4 on the
device (`rbd_simulate`), constant `torques` (the default control is `zero_torque!`), optional `externalwrenches`; `state.q`, `state.v` are advanced
in place.  Returns `(ts, qs, vs)` like the reference; `store = true` (the reference always stores, `ExpandingStorage`) keeps a host copy of q and v
(nq × B, nv × B) per step in `qs`, `vs` — one launch sequence per step; `store = false` (default here: B trajectories are B times the memory) runs
all the steps without returning to the host and returns `qs = vs = nothing`."""
function simulate(state::BatchedMechanismState{T}, final_time; Δt = 1e-4, torques = nothing, externalwrenches = nothing, stabilization_gains = :default,
        store::Bool = false) where {T}
    checkmodcount(state)
    nsteps = stepcount(T, final_time, Δt)
    B = batchsize(state)
    wext = densewrenches(state, externalwrenches)
    ts = range(zero(T), step = T(Δt), length = nsteps + 1)
    qs, vs = store ? ([hostcopy(state.q)], [hostcopy(state.v)]) : (nothing, nothing)
    for chunk in (store ? fill(1, nsteps) : (nsteps,))
        if size(state.s, 1) > 0     # contact points: the additional state is integrated beside (q, v) with the same tableau
            check(ccall((:rbd_simulate_contact, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Cdouble, Int32, Ref{RbdOpts}),
                state.ws, B, state.q, state.v, state.s, nullable(torques), nullable(wext), Float64(Δt), chunk, opts(state)), "rbd_simulate_contact")
        else
            check(ccall((:rbd_simulate, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Cdouble, Int32, Ref{RbdOpts}),
                state.ws, B, state.q, state.v, nullable(torques), nullable(wext), Float64(Δt), chunk,
                opts(state; stabilization = stabilization(state, stabilization_gains))), "rbd_simulate")
        end
        if store
            synchronize(state); push!(qs, hostcopy(state.q)); push!(vs, hostcopy(state.v))
        end
    end
    finish(state)
    ts, qs, vs
end

"""`simulate(state0, final_time, control!; Δt, stabilization_gains)` — src/simulate.jl:36-55 with ANY controller: `control!(torques, t, state)` is
called before every Runge-Kutta stage's `dynamics!` exactly as the reference's closure does (:42-48), with `torques` the nv × B buffer of the batch
(a `DeviceMatrix`: fill it with `copyto!` or a kernel of your own; the state must be device-resident — `ArgumentError` otherwise), `t` the stage time and `state` holding the STAGE state.
The stage arithmetic of MuntheKaasIntegrator.step (src/ode_integrators.jl:233-299) runs on the device (`rbd_mk_stage`: stage 0 snapshots the base
point, stages 1..3 take the previous stage's v̇, stage 4 closes the step); only the controller runs on the host.  Returns `(ts, qs, vs)` (`store`
as above).  Controllers that can run on the device should use `TorqueTable` / `PDControl` below: no host round trip per stage."""
function simulate(state::BatchedMechanismState{T}, final_time, control!::Function; Δt = 1e-4, externalwrenches = nothing, stabilization_gains = :default,
        store::Bool = false) where {T}
    checkmodcount(state)
    size(state.s, 1) == 0 || throw(ArgumentError("control! with contact points: drive dynamics! per stage yourself"))
    # rbd_mk_stage advances q, v in place on the device (it answers RBD_ERR_INVALID_ARGUMENT to host memory): say so here, by name
    state.memory == MEM_DEVICE || throw(ArgumentError("simulate with a control! callback needs a device-resident state (storage = :device); " *
                                                      "a host-resident one can be stepped with dynamics!(ẋ, result, state, x) and an integrator of your own"))
    nsteps = stepcount(T, final_time, Δt)
    B, nv, nc = batchsize(state), state.model.nv, state.model.nc
    wext = densewrenches(state, externalwrenches)
    o = opts(state; stabilization = stabilization(state, stabilization_gains))
    storage = Val(:device)
    τ, v̇, λ = newbuffer(storage, T, nv, B), newbuffer(storage, T, nv, B), newbuffer(storage, T, max(nc, 1), B)
    ts = range(zero(T), step = T(Δt), length = nsteps + 1)
    qs, vs = store ? ([hostcopy(state.q)], [hostcopy(state.v)]) : (nothing, nothing)
    stagetimes = (zero(T), T(Δt) / 2, T(Δt) / 2, T(Δt))     # runge_kutta_4's c (src/ode_integrators.jl:48-55)
    for k in 1:nsteps
        for stage in 0:4
            check(ccall((:rbd_mk_stage, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Int32, Cdouble, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
                state.ws, B, stage, Float64(Δt), state.q, state.v, stage > 0 ? pointer(v̇) : Ptr{T}(C_NULL), o), "rbd_mk_stage")
            stage == 4 && break
            synchronize(state)      # the controller may read the stage state from the host
            control!(τ, ts[k] + stagetimes[stage + 1], state)
            check(ccall((:rbd_dynamics, librbd_hip[]), Cint,
                (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
                state.ws, B, state.q, state.v, τ, nullable(wext), v̇, C_NULL, nc > 0 ? pointer(λ) : Ptr{T}(C_NULL), o), "rbd_dynamics")
        end
        if store
            synchronize(state); push!(qs, hostcopy(state.q)); push!(vs, hostcopy(state.v))
        end
    end
    finish(state)
    ts, qs, vs
end

"""`simulate(state0, final_time, control; Δt)` with a controller that runs on the device (no host round trip per Runge-Kutta stage — the
reference calls `control!(τ, t, state)` before every stage's `dynamics!`, src/simulate.jl:42-48):
`TorqueTable(τ, per_stage)` — τ :: (nv·B) × entries, entry 4·step + stage (the stage times) or entry `step` (zero-order hold);
`PDControl(kp, kd, q_des, τff)` — τ = τff − kp (q − q_des) − kd v on Revolute / Prismatic joints, on the stage state (kp, kd :: nv × 1)."""
struct TorqueTable{T}; torques::DeviceMatrix{T}; per_stage::Bool; end
struct PDControl{T}; kp::DeviceMatrix{T}; kd::DeviceMatrix{T}; q_des::Union{Nothing, DeviceMatrix{T}}; torques::Union{Nothing, DeviceMatrix{T}}; end
function simulate(state::BatchedMechanismState{T}, final_time, control::Union{TorqueTable{T}, PDControl{T}}; Δt = 1e-4, externalwrenches = nothing) where {T}
    checkmodcount(state)
    state.memory == MEM_DEVICE || throw(ArgumentError("device-side controllers need device-resident states (storage = :device)"))
    nsteps = stepcount(T, final_time, Δt)
    B, nq, nv = batchsize(state), state.model.nq, state.model.nv
    # the kernels index the controller's buffers without bounds: check them here (the C entry point sees only pointers)
    if control isa TorqueTable
        need = (control.per_stage ? 4 : 1) * nsteps
        size(control.torques, 1) == nv * B && size(control.torques, 2) >= need ||
            throw(DimensionMismatch("TorqueTable: need a (nv·B) × n table with n ≥ $need entries, got $(size(control.torques))"))
    else
        prod(size(control.kp)) == nv && prod(size(control.kd)) == nv || throw(DimensionMismatch("PDControl: kp, kd must hold nv = $nv gains"))
        control.q_des === nothing || size(control.q_des) == (nq, B) || throw(DimensionMismatch("PDControl: q_des must be nq × B"))
        control.torques === nothing || size(control.torques) == (nv, B) || throw(DimensionMismatch("PDControl: torques must be nv × B"))
    end
    wext = densewrenches(state, externalwrenches)
    vp(x) = x === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(x))
    ctl = control isa TorqueTable ? RbdControl(1, control.per_stage ? 1 : 0, vp(control.torques), C_NULL, C_NULL, C_NULL) :
                                    RbdControl(2, 0, vp(control.torques), vp(control.q_des), vp(control.kp), vp(control.kd))
    check(ccall((:rbd_simulate_controlled, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ref{RbdControl}, Ptr{T}, Cdouble, Int32, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, ctl, nullable(wext), Float64(Δt), nsteps, opts(state)), "rbd_simulate_controlled")
    finish(state)
    range(zero(T), step = T(Δt), length = nsteps + 1), nothing, nothing
end

# ---- kinematics by-products of the same forward-kinematics pass (root frame) -------------------------------------------------------
# momentum_matrix!(out, state) mechanism_algorithms.jl:313-327: out is (6·nv) × B, per state a 6 × nv column-major matrix
function momentum_matrix!(out::Buffer{T}, state::BatchedMechanismState{T}) where {T}
    checkmodcount(state)
    B = batchsize(state)
    size(out) == (6 * state.model.nv, B) || throw(DimensionMismatch())
    check(ccall((:rbd_kinematics, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, out, C_NULL, C_NULL, opts(state)), "rbd_kinematics")
    finish(state)
    out
end

# center_of_mass(state) :28-50 -> 3 × B;  kinetic_energy / gravitational_potential_energy mechanism_state.jl:886-903 -> B each
function kin(state::BatchedMechanismState{T}, n::Int, slot::Int) where {T}
    B = batchsize(state)
    out = newbuffer(Val(state.memory == MEM_HOST ? :host : :device), T, n, B)
    ptrs = (C_NULL, C_NULL, C_NULL)
    check(ccall((:rbd_kinematics, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.v, C_NULL, slot == 2 ? pointer(out) : Ptr{T}(C_NULL), slot == 3 ? pointer(out) : Ptr{T}(C_NULL), opts(state)), "rbd_kinematics")
    synchronize(state)
    out isa Matrix ? out : Array(out)
end
center_of_mass(state::BatchedMechanismState) = kin(state, 3, 2)
kinetic_energy(state::BatchedMechanismState) = kin(state, 2, 3)[1, :]
gravitational_potential_energy(state::BatchedMechanismState) = kin(state, 2, 3)[2, :]

# momentum(state), momentum_rate_bias(state) mechanism_state.jl:975-987 -> 6 × B each
function momenta(state::BatchedMechanismState{T}) where {T}
    B = batchsize(state)
    out = newbuffer(Val(state.memory == MEM_HOST ? :host : :device), T, 12, B)
    check(ccall((:rbd_momentum, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Ptr{T}, Ptr{T}, Ref{RbdOpts}), state.ws, B, state.q, state.v, out, opts(state)),
        "rbd_momentum")
    synchronize(state)
    out isa Matrix ? out : Array(out)
end
momentum(state::BatchedMechanismState) = momenta(state)[1:6, :]
momentum_rate_bias(state::BatchedMechanismState) = momenta(state)[7:12, :]

# geometric_jacobian!(out, state, path) :80-99 in the root frame; the path is given by its end bodies (path(mechanism, base, body))
function geometric_jacobian!(out::Buffer{T}, state::BatchedMechanismState{T}, base::RigidBody, body::RigidBody) where {T}
    checkmodcount(state)
    B = batchsize(state)
    size(out) == (6 * state.model.nv, B) || throw(DimensionMismatch())
    check(ccall((:rbd_geometric_jacobian, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{T}, Int32, Int32, Ptr{T}, Ref{RbdOpts}),
        state.ws, B, state.q, state.model.bodyindex[base], state.model.bodyindex[body], out, opts(state)), "rbd_geometric_jacobian")
    finish(state)
    out
end

# ---- multi-GPU: one process per GPU, the batch sharded by state, v̇ gathered over RCCL (SURVEY.md §8 e) ------------------------------
mutable struct RbdComm
    handle::Ptr{Cvoid}
    world::Int; rank::Int
end
"128-byte unique id (call on one rank, hand to the others by whatever means the launcher offers — MPI, a file, a socket)"
function unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:rbd_comm_unique_id, librbd_hip[]), Cint, (Ptr{Cvoid},), id), "rbd_comm_unique_id")
    id
end
function RbdComm(id::Vector{UInt8}, world::Integer, rank::Integer; device::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rbd_comm_create, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}), id, world, rank, device, h), "rbd_comm_create")
    c = RbdComm(h[], world, rank)
    finalizer(x -> ccall((:rbd_comm_destroy, librbd_hip[]), Cint, (Ptr{Cvoid},), x.handle), c)
    c
end
"gathered (n × world·B_local) ← every rank's shard (n × B_local); `root = nothing`: on every rank, else on that rank only"
function gather!(gathered::Union{DeviceMatrix{T}, Nothing}, comm::RbdComm, shard::DeviceMatrix{T}; root = nothing) where {T}
    check(ccall((:rbd_gather, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32, Ptr{Cvoid}),
        comm.handle, T === Float64 ? 0 : 1, shard.ptr, gathered === nothing ? C_NULL : gathered.ptr, prod(size(shard)), root === nothing ? -1 : root, C_NULL), "rbd_gather")
    gathered
end
"... of shards with different numbers of states: `cols[r + 1]` states on rank r (a batch that does not divide by the number of ranks), gathered n × sum(cols)"
function gatherv!(gathered::Union{DeviceMatrix{T}, Nothing}, comm::RbdComm, shard::DeviceMatrix{T}, cols::AbstractVector{<:Integer}; root = nothing) where {T}
    length(cols) == comm.world && cols[comm.rank + 1] == size(shard, 2) || throw(DimensionMismatch("cols must list every rank's states; cols[rank + 1] must be this shard's"))
    counts = Int64[size(shard, 1) * c for c in cols]
    check(ccall((:rbd_gatherv, librbd_hip[]), Cint, (Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Int32, Ptr{Cvoid}),
        comm.handle, T === Float64 ? 0 : 1, shard.ptr, gathered === nothing ? C_NULL : gathered.ptr, counts, root === nothing ? -1 : root, C_NULL), "rbd_gatherv")
    gathered
end

end # module
