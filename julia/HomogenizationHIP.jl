# HomogenizationHIP.jl -- Julia host binding of libhmg_hip.so (C ABI: include/hmg.h) for haampie/Homogenization.jl.
#
# The reference's seam for the hot path is the matrix type of `LevelState{T,Tv<:AbstractMatrix{T}}`
# (src/multigrid.jl:7-13) and the methods dispatched on it.  This module adds ONE matrix type backed by device memory
# (`HipMatrix`) and forwards every hot-path method to the library with `ccall`; no CUDA.jl / AMDGPU.jl, no kernel DSL.
# It is kept line for line parallel to the executable Python mirror of this repository
# (homogenization.jl_amd/api.py = the method layer, driver.py = the checkerboard_homogenization loop, dist.py = the
# multi-GPU layer), whose ctypes tests stand in for it: the build image has no Julia toolchain, so THIS FILE HAS NEVER
# BEEN EXECUTED.  Every function names the reference method it replaces (file:line in the reference checkout).
module HomogenizationHIP

using Homogenization
using Homogenization: Mesh, Tets64, Tris64, Tet64, Tri64, ImplicitFineGrid, LevelState, L2PlusDivAGrad, SimpleDiffusion,
                      ZeroDirichletConstraint, hypercube, nelements, nnodes, base_mesh, nlevels, refined_mesh
import Homogenization: mul!, local_residual!, apply_constraint!, broadcast_interfaces!, zero_out_all_but_one!,
                       restrict_to!, interpolate_and_sum_to!, smoothing_steps!, vcycle!, copy_to_base!, distribute!,
                       local_rhs!, rhs_aξ∇v!, next_rhs!, integrate_first_term, integrate_terms, integrate_area,
                       checkerboard_homogenization
import LinearAlgebra: dot, axpy!, norm
import Random: rand!
using StaticArrays: SVector

const LIB = get(ENV, "HMG_LIB", "libhmg_hip.so")

check(rc) = rc == 0 || error(unsafe_string(ccall((:hmg_last_error, LIB), Cstring, ())))

# ---- context: one GPU + one HIP stream (api.Context) -----------------------------------------------------------------
mutable struct HipContext
    h::Ptr{Cvoid}
    function HipContext(device::Integer = 0)
        h = Ref{Ptr{Cvoid}}()
        check(ccall((:hmg_ctx_create, LIB), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h))
        finalizer(c -> ccall((:hmg_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.h), new(h[]))
    end
    HipContext(h::Ptr{Cvoid}, ::Val{:handle}) = new(h)                   # (an existing handle: see the stream constructor below)
end
sync(c::HipContext) = check(ccall((:hmg_ctx_sync, LIB), Cint, (Ptr{Cvoid},), c.h))
release_memory(c::HipContext) = check(ccall((:hmg_ctx_release_memory, LIB), Cint, (Ptr{Cvoid},), c.h))
set_option!(c::HipContext, name::String, v::Integer) =
    check(ccall((:hmg_ctx_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), c.h, name, v))

# in-library communicator (RCCL over xGMI, one Julia process per GPU; api.Context.comm_init).  `bcast` hands rank 0's
# 128 bytes to every rank, e.g. `id -> MPI.Bcast!(id, 0, comm)` with MPI.jl.
function comm_init!(c::HipContext, nranks::Integer, rank::Integer, bcast)
    id = zeros(UInt8, 128)
    rank == 0 && check(ccall((:hmg_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    bcast(id)
    check(ccall((:hmg_comm_init, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), c.h, nranks, rank, id))
end
function comm_sum(c::HipContext, vals::Float64...)                       # the driver's per-cycle integrals over ranks
    a = collect(vals)
    check(ccall((:hmg_comm_sum_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), c.h, a, length(a)))
    a
end

# ---- grid: ImplicitFineGrid(base, levels) + ZeroDirichletConstraint (src/implicit_fine_grid.jl:13-18, :80-84) --------
# The library derives the reference tables, the interface / boundary maps and the partition itself.
mutable struct HipGrid{dim}
    h::Ptr{Cvoid}
    ctx::HipContext
    implicit::ImplicitFineGrid       # the reference object the scripts pass around (its tables are not used)
    levels::Int
    bound::Any                       # the operator whose sigma / lambda the library currently holds
    bound_lambda::Float64
end

flat_nodes(base::Mesh{dim}) where {dim} = collect(reinterpret(Float64, base.nodes))          # dim * nnodes, node-major
flat_cells(base::Mesh) = collect(Int64, Iterators.flatten(base.elements))                    # 1-based, ascending tuples

function HipGrid(ctx::HipContext, implicit::ImplicitFineGrid{dim}) where {dim}
    base = base_mesh(implicit)
    coords, cells = flat_nodes(base), flat_cells(base)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_grid_create, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ref{Ptr{Cvoid}}),
                ctx.h, dim, nlevels(implicit), nnodes(base), coords, nelements(base), cells, h))
    finalizer(g -> ccall((:hmg_grid_destroy, LIB), Cint, (Ptr{Cvoid},), g.h),
              HipGrid{dim}(h[], ctx, implicit, nlevels(implicit), nothing, NaN))
end

# one rank's share of a global base mesh (dist.PartitionedGrid): owner[c] = rank that owns coarse cell c
function HipGrid(ctx::HipContext, implicit::ImplicitFineGrid{dim}, owner::Vector{Int32}, rank::Integer, nranks::Integer) where {dim}
    base = base_mesh(implicit)
    coords, cells = flat_nodes(base), flat_cells(base)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_grid_create_partition, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int32}, Cint, Cint, Ref{Ptr{Cvoid}}),
                ctx.h, dim, nlevels(implicit), nnodes(base), coords, nelements(base), cells, owner, rank, nranks, h))
    g = finalizer(g -> ccall((:hmg_grid_destroy, LIB), Cint, (Ptr{Cvoid},), g.h),
                  HipGrid{dim}(h[], ctx, implicit, nlevels(implicit), nothing, NaN))
    check(ccall((:hmg_grid_use_comm, LIB), Cint, (Ptr{Cvoid},), g.h))     # exchange through the context's communicator
    g
end

ncells(g::HipGrid) = Int(ccall((:hmg_grid_ncells, LIB), Int64, (Ptr{Cvoid},), g.h))
nf(g::HipGrid, level) = Int(ccall((:hmg_grid_nf, LIB), Int64, (Ptr{Cvoid}, Cint), g.h, level))

# The reference passes the operator to every call and mutates `ops.λ` in place between outer steps
# (src/examples/homogenized_coefficients.jl:330-333); the library keeps ONE operator per grid, so every forwarded
# method re-binds when the operator object or its λ differs from what the library holds (api._Operator._bind).
sigmas(A::L2PlusDivAGrad) = collect(reinterpret(Float64, A.σs))                               # dim * ncells
sigmas(A::SimpleDiffusion{dim}, ne) where {dim} = fill(Float64(A.a), dim * ne)
lambda(A::L2PlusDivAGrad) = Float64(A.λ)
lambda(::SimpleDiffusion) = 0.0
function bind!(g::HipGrid, A)
    if g.bound !== A
        s = A isa SimpleDiffusion ? sigmas(A, ncells(g)) : sigmas(A)
        check(ccall((:hmg_grid_set_operator, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64), g.h, s, lambda(A)))
        g.bound, g.bound_lambda = A, lambda(A)
    elseif g.bound_lambda != lambda(A)
        check(ccall((:hmg_grid_set_lambda, LIB), Cint, (Ptr{Cvoid}, Float64), g.h, lambda(A)))
        g.bound_lambda = lambda(A)
    end
    g
end
bind!(g::HipGrid, ops::AbstractVector, k::Int) = bind!(g, ops[k])

# domain shrink to a prefix of cells / nodes + new Dirichlet boundary: replaces rebuilding ImplicitFineGrid and slicing
# every LevelState (shrink_level_state, src/examples/homogenized_coefficients.jl:54-60, :309-336) -- level vectors
# keep their storage, their columns are a prefix
shrink!(g::HipGrid, ncells_prefix::Integer, nnodes_prefix::Integer) =
    check(ccall((:hmg_grid_shrink, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), g.h, ncells_prefix, nnodes_prefix))

# The sixth finest-level vector of the library's default V-cycle form (include/hmg.h: hmg_grid_reserve_spare).  true: reserve it now
# (an error if the device memory is not there); false: release it and keep the reference's five-vector footprint.  Without a call
# it is reserved when the first HipMatrix of the finest level is created.
reserve_spare!(g::HipGrid, enable::Bool = true) =
    check(ccall((:hmg_grid_reserve_spare, LIB), Cint, (Ptr{Cvoid}, Cint), g.h, enable ? 1 : 0))

# ---- HipMatrix: the AbstractMatrix a LevelState is parametrised with (api.DeviceMatrix) ------------------------------
mutable struct HipMatrix <: AbstractMatrix{Float64}
    h::Ptr{Cvoid}
    grid::HipGrid
    level::Int
end
function HipMatrix(g::HipGrid, level::Int)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_vec_create, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), g.h, level, h))
    finalizer(v -> ccall((:hmg_vec_destroy, LIB), Cint, (Ptr{Cvoid},), v.h), HipMatrix(h[], g, level))
end
Base.size(v::HipMatrix) = (nf(v.grid, v.level), ncells(v.grid))
Base.getindex(::HipMatrix, ::Int...) = error("HipMatrix lives in HBM: use Array(v) (Nf x Ne, the reference's hierarchical node order)")
Base.similar(v::HipMatrix) = HipMatrix(v.grid, v.level)
Base.copy(v::HipMatrix) = copyto!(similar(v), v)
Base.copyto!(v::HipMatrix, a::Matrix{Float64}) =
    (check(ccall((:hmg_vec_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), v.h, a)); v)
Base.Array(v::HipMatrix) =
    (a = Matrix{Float64}(undef, size(v)...); check(ccall((:hmg_vec_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), v.h, a)); a)
Base.fill!(v::HipMatrix, x) = (check(ccall((:hmg_vec_fill, LIB), Cint, (Ptr{Cvoid}, Float64), v.h, x)); v)
Base.copyto!(d::HipMatrix, s::HipMatrix) = (check(ccall((:hmg_vec_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), d.h, s.h)); d)
# rand!(x) (src/examples/homogenized_coefficients.jl:246): seeded and layout independent on the device
rand!(v::HipMatrix; seed::Integer = 1, cell_offset::Integer = 0) =
    (check(ccall((:hmg_vec_fill_random, LIB), Cint, (Ptr{Cvoid}, UInt64, Int64), v.h, seed, cell_offset)); v)
# BLAS-1 over the raw storage: shared DOFs counted once per copy, like BLAS on the reference's Matrix (src/multigrid.jl:54-68)
dot(x::HipMatrix, y::HipMatrix) =
    (o = Ref(0.0); check(ccall((:hmg_vec_dot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), x.h, y.h, o)); o[])
norm(x::HipMatrix) = sqrt(dot(x, x))
axpy!(a, x::HipMatrix, y::HipMatrix) =
    (check(ccall((:hmg_vec_axpy, LIB), Cint, (Float64, Ptr{Cvoid}, Ptr{Cvoid}), a, x.h, y.h)); y)
# `curr.p .= curr.r .+ c .* curr.p` (src/multigrid.jl:68)
xpby!(r::HipMatrix, c, p::HipMatrix) =
    (check(ccall((:hmg_vec_xpby, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), r.h, c, p.h)); p)
# norm(r) after zero_out_all_but_one!(r), without destroying r (src/examples/homogenized_coefficients.jl:286-287)
norm_unique(r::HipMatrix) =
    (o = Ref(0.0); check(ccall((:hmg_vec_norm_unique, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), r.h, o)); o[])

# LevelState(ncells, nnodes, Float64) uses zeros(...) (src/multigrid.jl:18-25): constructed explicitly here
LevelState(g::HipGrid, level::Int) = LevelState{Float64,HipMatrix}((HipMatrix(g, level) for _ in 1:5)...)
const HipState = LevelState{Float64,HipMatrix}
handles(levels::Vector{HipState}) = Ptr{Cvoid}[getfield(l, f).h for l in levels for f in (:x, :b, :r, :p, :Ap)]

# ---- hot-path methods ------------------------------------------------------------------------------------------------
# mul!(α, base, A, x, y): y += α A x (src/apply_local_operators.jl:40-72, :85-133)
mul!(α::Float64, base::Mesh, A::Union{L2PlusDivAGrad,SimpleDiffusion}, x::HipMatrix, y::HipMatrix) =
    (bind!(x.grid, A); check(ccall((:hmg_apply, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Ptr{Cvoid}), x.grid.h, x.level, α, x.h, y.h)); y)
# local_residual!: r = b - A x, constraint (src/apply_local_operators.jl:7-27)
local_residual!(implicit, A::Union{L2PlusDivAGrad,SimpleDiffusion}, c::HipState, k::Int) =
    (bind!(c.x.grid, A); check(ccall((:hmg_residual, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), c.x.grid.h, k, c.x.h, c.b.h, c.r.h)))
# typed on Matrix / Vector in the reference (src/implicit_fine_grid.jl:94,148,178): additional methods, nothing changes there
apply_constraint!(x::HipMatrix, level::Int, z, implicit) =
    (check(ccall((:hmg_constraint, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), x.grid.h, level, x.h)); x)
broadcast_interfaces!(x::HipMatrix, implicit, level::Int) =                    # src/implicit_fine_grid.jl:209-328
    (check(ccall((:hmg_interface_sum, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), x.grid.h, level, x.h)); x)
zero_out_all_but_one!(x::HipMatrix, implicit, level::Int) =                    # src/implicit_fine_grid.jl:334-386
    (check(ccall((:hmg_zero_duplicates, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), x.grid.h, level, x.h)); x)
restrict_to!(y::HipMatrix, P, x::HipMatrix) =                                  # src/interpolation.jl:52-62
    (check(ccall((:hmg_restrict, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}), x.grid.h, x.level, x.h, y.h)); y)
interpolate_and_sum_to!(y::HipMatrix, P, x::HipMatrix) =                       # src/interpolation.jl:64-74
    (check(ccall((:hmg_prolong_add, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}), y.grid.h, y.level, x.h, y.h)); y)
copy_to_base!(u::Vector{Float64}, v::HipMatrix, implicit) =                    # src/implicit_fine_grid.jl:148-172
    (check(ccall((:hmg_gather_base, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), v.grid.h, v.h, u)); u)
distribute!(v::HipMatrix, u::Vector{Float64}, implicit) =                      # src/implicit_fine_grid.jl:178-202
    (check(ccall((:hmg_scatter_base, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), v.grid.h, u, v.h)); v)

# level-1 solve: BaseLevel(Float64, cholesky(assemble_checkerboard(...)[interior, interior]), ...) is replaced by the
# library's device-resident Chebyshev-preconditioned CG on the same matrix (src/examples/homogenized_coefficients.jl:259-261,
# src/multigrid.jl:30-41, :74-93).  Construct one after every change of sigma, lambda or the domain.
struct HipBaseLevel
    grid::HipGrid
    HipBaseLevel(g::HipGrid, A) = (bind!(g, A); check(ccall((:hmg_coarse_setup, LIB), Cint, (Ptr{Cvoid},), g.h)); new(g))
end
# iterations of the last level-1 solve (waits for it); an unconverged solve is an error, as it is for the call that
# synchronises behind its V-cycle (include/hmg.h, hmg_coarse_last_iterations)
function last_iterations(b::HipBaseLevel)
    n = Int(ccall((:hmg_coarse_last_iterations, LIB), Cint, (Ptr{Cvoid},), b.grid.h))
    n < 0 && error(unsafe_string(ccall((:hmg_last_error, LIB), Cstring, ())))
    n
end
coarse_misses(b::HipBaseLevel) = Int(ccall((:hmg_coarse_misses, LIB), Int64, (Ptr{Cvoid},), b.grid.h))

# fused fast path: one call per smoother / per V-cycle (src/multigrid.jl:46-119).  steps_coarse = 2 because the reference
# does not forward `steps` to the recursive call (src/multigrid.jl:109).
smoothing_steps!(steps::Integer, implicit, ops, c::HipState, k::Int) =
    (bind!(c.x.grid, ops); check(ccall((:hmg_smooth, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                                        c.x.grid.h, k, steps, c.x.h, c.b.h, c.r.h, c.p.h, c.Ap.h)))
function vcycle!(implicit, base::HipBaseLevel, ops::Vector, levels::Vector{HipState}, k::Int, steps = 2)
    g = levels[1].x.grid
    bind!(g, ops, k)
    check(ccall((:hmg_vcycle, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}), g.h, k, steps, 2, handles(levels)))
    nothing
end

# Optional, once after the level states exist and before they hold data: which memory block plays x, b, r, p, Ap of
# level k is chosen by timing that level's share of a V-cycle (include/hmg.h, hmg_level_tune_placement).
function tune_placement!(ops::Vector, levels::Vector{HipState}, k::Int, steps = 3; trials = 8, extra = 2)
    g = levels[1].x.grid
    bind!(g, ops, k)
    ms = zeros(Float64, 2)
    check(ccall((:hmg_level_tune_placement, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}, Cint, Cint, Ptr{Float64}),
                g.h, k, steps, handles(levels), extra, trials, ms))
    (ms[1], ms[2])
end

# ---- driver right-hand sides and integrals (src/examples/homogenized_coefficients.jl:449-474, :592-713; src/implicit_fine_grid.jl:391-409)
rhs_aξ∇v!(b::HipMatrix, ∂ϕ∂xᵢs, implicit, σs, ξ::SVector{dim,Float64}) where {dim} =
    (check(ccall((:hmg_rhs_axi_grad, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), b.grid.h, collect(ξ), b.h)); b)
local_rhs!(b::HipMatrix, implicit) = (check(ccall((:hmg_local_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), b.grid.h, b.h)); b)
next_rhs!(b::HipMatrix, x::HipMatrix, implicit, ops::L2PlusDivAGrad) =
    (bind!(x.grid, ops); check(ccall((:hmg_next_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), x.grid.h, x.h, b.h)); nothing)
function integrate(g::HipGrid, mode::Integer, v::HipMatrix, second, nsubset::Integer)
    o = Ref(0.0)
    check(ccall((:hmg_integrate, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Ptr{Float64}, Ref{Float64}),
                g.h, mode, v.h, second === nothing ? C_NULL : second.h, nsubset, C_NULL, o))
    o[]
end
# subsets are prefixes 1:n of the ∞-norm ordered cells (find_elements_in_radius, :32-43).  The first term's dot(∂ϕ, P) is
# the entry of rhs_aξ∇v! for the same ξ: the caller's right-hand side b of outer step 0 is passed along.
integrate_first_term(v₀::HipMatrix, b::HipMatrix, subset::AbstractUnitRange, ops::L2PlusDivAGrad) =
    (bind!(v₀.grid, ops); integrate(v₀.grid, 0, v₀, b, length(subset)))
integrate_terms(vₖ::HipMatrix, vₖ₋₁::HipMatrix, implicit, subset::AbstractUnitRange, ops::L2PlusDivAGrad) =
    (bind!(vₖ.grid, ops); integrate(vₖ.grid, 1, vₖ, vₖ₋₁, length(subset)))
integrate_area(g::HipGrid, v::HipMatrix, subset::AbstractUnitRange) = integrate(g, 2, v, nothing, length(subset))

# ---- checkerboard_homogenization(n, ElT; refinements, smoothing_steps, tolerance, ξ, save, backend = :hip) -----------
# The reference's loop (src/examples/homogenized_coefficients.jl:174-343) with HipMatrix states; mirrors driver.py.
function checkerboard_homogenization(n::Int, elementtype::Type{<:Union{Tri64,Tet64}}, ::Val{:hip};
                                     refinements::Int = 2, smoothing_steps::Int = 3, tolerance::Float64 = 1e-4,
                                     ξ = nothing, ctx::HipContext = HipContext(0), seed::Integer = 0)
    H = Homogenization
    dim = elementtype <: Tet64 ? 3 : 2
    ξ = ξ === nothing ? SVector{dim,Float64}(ntuple(_ -> 1 / sqrt(dim), dim)) : ξ
    λ, σ = 1.0, 0.0
    box_radius = H.compute_box_radius(0, n)
    boundary_layer = H.compute_boundary_layer(λ, n)
    total_radius = box_radius + boundary_layer
    width = 2 * total_radius
    base = H.order_nodes_and_elements_by_magnitude(hypercube(elementtype, width, origin = ntuple(_ -> -Float64(total_radius), dim)))
    cond = H.conductivity_per_element(base, H.generate_conductivity(base, width), ntuple(_ -> total_radius + 1.0, dim))
    total_grids = refinements + 1
    implicit = ImplicitFineGrid(base, total_grids)              # reference object (host tables); the device grid below
    g = HipGrid(ctx, implicit)
    nodes, edges, faces = H.list_boundary_nodes_edges_faces(base)
    constraint = ZeroDirichletConstraint(nodes, edges, faces)
    diff_terms = H.build_local_diffusion_operators(implicit.reference)
    mass_terms = H.build_local_mass_matrices(implicit.reference)
    ops = [L2PlusDivAGrad(d, m, constraint, λ, cond) for (d, m) in zip(diff_terms, mass_terms)]
    states = [LevelState(g, k) for k in 1:total_grids]
    top = states[end]
    rand!(top.x; seed = seed + 1)
    broadcast_interfaces!(top.x, implicit, total_grids)
    apply_constraint!(top.x, total_grids, constraint, implicit)
    rhs_aξ∇v!(top.b, nothing, implicit, cond, ξ)
    v_prev = similar(top.x)
    cur = base
    for k in 0:n
        base_level = HipBaseLevel(g, ops[end])                  # level-1 operator for the current λ / domain
        Δσ, Δσ_prev = 0.0, 0.0
        for i in 1:1000
            vcycle!(implicit, base_level, ops, states, total_grids, smoothing_steps)
            interior = 1:H.find_elements_in_radius(cur, box_radius)
            area = integrate_area(g, top.x, interior)
            integral = k == 0 ? integrate_first_term(top.x, top.b, interior, ops[end]) :
                                integrate_terms(top.x, v_prev, implicit, interior, ops[end])
            Δσ = 2.0^k * integral / area
            @info "Next Δσ" k i norm_unique(top.r) (σ + Δσ) abs(Δσ - Δσ_prev)
            abs(Δσ - Δσ_prev) < tolerance && break
            Δσ_prev = Δσ
        end
        σ += Δσ
        λ /= 2
        box_radius = H.compute_box_radius(k + 1, n)
        boundary_layer = H.compute_boundary_layer(λ, n)
        box_radius + boundary_layer > total_radius && break
        total_radius = box_radius + boundary_layer
        nn_keep = H.find_nodes_in_radius(cur, total_radius)
        ne_keep = H.find_elements_in_radius(cur, total_radius)
        cur = Mesh(cur.nodes[1:nn_keep], cur.elements[1:ne_keep])
        shrink!(g, ne_keep, nn_keep)                            # new boundary; level vectors keep their storage
        apply_constraint!(top.x, total_grids, constraint, implicit)
        copyto!(v_prev, top.x)
        for op in ops
            op.λ = λ                                            # (the next forwarded call re-binds, see bind!)
        end
        next_rhs!(top.b, top.x, implicit, ops[end])
    end
    σ
end
checkerboard_homogenization(n::Int, ElT; backend::Symbol = :cpu, kwargs...) =
    backend == :hip ? checkerboard_homogenization(n, ElT, Val(:hip); kwargs...) :
                      invoke(checkerboard_homogenization, Tuple{Int,Type{<:Homogenization.ElementType}}, n, ElT; kwargs...)


# ---- the rest of the C ABI (include/hmg.h): everything the executable Python mirror binds is bound here too -----------
# tests/test_binding_drift.py holds every `ccall` of this file against the header: name, arity, type class of every argument.
version() = Int(ccall((:hmg_version, LIB), Cint, ()))

# context: a caller-owned HIP stream, float options, counters, the apply timer, the scalar bank (api.Context)
function HipContext(device::Integer, stream::Ptr{Cvoid})                 # kernels are enqueued on the caller's stream
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_ctx_create_on_stream, LIB), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, h))
    finalizer(c -> ccall((:hmg_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.h), HipContext(h[], Val(:handle)))
end
set_option!(c::HipContext, name::String, v::Float64) =
    check(ccall((:hmg_ctx_set_option_f64, LIB), Cint, (Ptr{Cvoid}, Cstring, Float64), c.h, name, v))
counter(c::HipContext, name::String) = Int(ccall((:hmg_ctx_counter, LIB), Int64, (Ptr{Cvoid}, Cstring), c.h, name))
stream(c::HipContext) = ccall((:hmg_ctx_stream, LIB), Ptr{Cvoid}, (Ptr{Cvoid},), c.h)
scalar_bank(c::HipContext) = ccall((:hmg_ctx_scalar_bank, LIB), Ptr{Cvoid}, (Ptr{Cvoid},), c.h)
set_scalar_bank!(c::HipContext, device_doubles16::Ptr{Cvoid}) =
    check(ccall((:hmg_ctx_set_scalar_bank, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), c.h, device_doubles16))
function apply_timing(c::HipContext)                                     # (launches, ms, algorithmic bytes) since "time_apply" was set
    n, ms, by = Ref{Int64}(0), Ref{Float64}(0.0), Ref{Float64}(0.0)
    check(ccall((:hmg_ctx_apply_timing, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Float64}, Ref{Float64}), c.h, n, ms, by))
    (n[], ms[], by[])
end
function apply_timing(c::HipContext, level::Integer)                     # ... of one level
    n, ms, by = Ref{Int64}(0), Ref{Float64}(0.0), Ref{Float64}(0.0)
    check(ccall((:hmg_ctx_apply_timing_level, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}, Ref{Float64}, Ref{Float64}),
                c.h, level, n, ms, by))
    (n[], ms[], by[])
end
comm_destroy!(c::HipContext) = check(ccall((:hmg_comm_destroy, LIB), Cint, (Ptr{Cvoid},), c.h))
function comm_stats(c::HipContext)                                       # (collectives issued, doubles moved)
    n, d = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:hmg_comm_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), c.h, n, d))
    (n[], d[])
end

# grid queries and tables (api.ImplicitFineGrid)
nnodes_base(g::HipGrid) = Int(ccall((:hmg_grid_nnodes, LIB), Int64, (Ptr{Cvoid},), g.h))
nlevels_grid(g::HipGrid) = Int(ccall((:hmg_grid_nlevels, LIB), Cint, (Ptr{Cvoid},), g.h))
ld(g::HipGrid, level) = Int(ccall((:hmg_grid_ld, LIB), Int64, (Ptr{Cvoid}, Cint), g.h, level))
function table_i32(g::HipGrid, which::String, level::Integer = 1)
    n = Ref{Int64}(0)
    check(ccall((:hmg_grid_table_i32, LIB), Cint, (Ptr{Cvoid}, Cint, Cstring, Ptr{Int32}, Int64, Ref{Int64}), g.h, level, which, C_NULL, 0, n))
    out = zeros(Int32, n[])
    check(ccall((:hmg_grid_table_i32, LIB), Cint, (Ptr{Cvoid}, Cint, Cstring, Ptr{Int32}, Int64, Ref{Int64}), g.h, level, which, out, n[], n))
    out
end
function table_f64(g::HipGrid, which::String, level::Integer = 1)
    n = Ref{Int64}(0)
    check(ccall((:hmg_grid_table_f64, LIB), Cint, (Ptr{Cvoid}, Cint, Cstring, Ptr{Float64}, Int64, Ref{Int64}), g.h, level, which, C_NULL, 0, n))
    out = zeros(Float64, n[])
    check(ccall((:hmg_grid_table_f64, LIB), Cint, (Ptr{Cvoid}, Cint, Cstring, Ptr{Float64}, Int64, Ref{Int64}), g.h, level, which, out, n[], n))
    out
end

# level vectors over caller-owned device memory (api.DeviceMatrix.wrap): ld(g, level) * ncells(g) doubles
function HipMatrix(g::HipGrid, level::Int, device_ptr::Ptr{Cvoid})
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_vec_wrap, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), g.h, level, device_ptr, h))
    finalizer(v -> ccall((:hmg_vec_destroy, LIB), Cint, (Ptr{Cvoid},), v.h), HipMatrix(h[], g, level))
end
device_ptr(v::HipMatrix) = ccall((:hmg_vec_device_ptr, LIB), Ptr{Cvoid}, (Ptr{Cvoid},), v.h)

# out = (src or 0) + alpha A x, optionally constrained, in one kernel (api.apply_ex); the two halves of a V-cycle level
# (api.vcycle_down / vcycle_up: src/multigrid.jl:100-106 and :112-115); the level-1 solve alone (src/multigrid.jl:74-93)
function apply_ex!(out::HipMatrix, alpha, ops, x::HipMatrix, src::Union{HipMatrix,Nothing} = nothing; constrain::Bool = false)
    bind!(x.grid, ops)
    check(ccall((:hmg_apply_ex, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint),
                x.grid.h, x.level, alpha, x.h, src === nothing ? C_NULL : src.h, out.h, constrain ? 1 : 0))
    out
end
function vcycle_down!(ops::Vector, levels::Vector{HipState}, k::Int, steps = 2)
    g = levels[k].x.grid
    bind!(g, ops, k)
    check(ccall((:hmg_vcycle_down, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}), g.h, k, steps, handles(levels)))
end
function vcycle_up!(ops::Vector, levels::Vector{HipState}, k::Int, steps = 2)
    g = levels[k].x.grid
    bind!(g, ops, k)
    check(ccall((:hmg_vcycle_up, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Ptr{Cvoid}}), g.h, k, steps, handles(levels)))
end
coarse_solve!(x1::HipMatrix, b1::HipMatrix) =
    check(ccall((:hmg_coarse_solve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), x1.grid.h, b1.h, x1.h))

# multi-GPU hooks for hosts that bring their own transport (dist.Exchange with backend "torch"; MPI.jl in Julia): the cut
# tables of a host-side partition analysis, the callbacks of the sum over ranks (C function pointers: @cfunction), buffer
# sizes and the message list of the sharers-only exchange.  The in-library RCCL communicator needs none of them
# (comm_init! + the partitioned HipGrid constructor above).
function set_cut!(g::HipGrid, nglobal::NTuple{3,Int}, face_gid::Vector{Int64}, face_cell_lid::Vector{Int32},
                  edge_gid::Vector{Int64}, edge_cell_lid::Vector{Int32}, node_gid::Vector{Int64}, node_cell_lid::Vector{Int32})
    check(ccall((:hmg_grid_set_cut, LIB), Cint,
                (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Int64, Ptr{Int64}, Ptr{Int32}, Int64, Ptr{Int64}, Ptr{Int32}),
                g.h, nglobal[1], nglobal[2], nglobal[3], length(face_gid), face_gid, face_cell_lid, length(edge_gid), edge_gid,
                edge_cell_lid, length(node_gid), node_gid, node_cell_lid))
end
set_exchange!(g::HipGrid, exchange::Ptr{Cvoid}, scalar_sum::Ptr{Cvoid}, user::Ptr{Cvoid}, device_buf::Ptr{Cvoid}, ndoubles::Integer) =
    check(ccall((:hmg_grid_set_exchange, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                g.h, exchange, scalar_sum, user, device_buf, ndoubles))
set_exchange_async!(g::HipGrid, begin_fn::Ptr{Cvoid}, end_fn::Ptr{Cvoid}) =
    check(ccall((:hmg_grid_set_exchange_async, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), g.h, begin_fn, end_fn))
set_exchange_p2p!(g::HipGrid, enabled::Bool, p2p::Ptr{Cvoid}, p2p_begin::Ptr{Cvoid}, device_stage::Ptr{Cvoid}, ndoubles::Integer) =
    check(ccall((:hmg_grid_set_exchange_p2p, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64),
                g.h, enabled ? 1 : 0, p2p, p2p_begin, device_stage, ndoubles))
set_overlap!(g::HipGrid, enabled::Bool) = check(ccall((:hmg_grid_set_overlap, LIB), Cint, (Ptr{Cvoid}, Cint), g.h, enabled ? 1 : 0))
cut_buffer_doubles(g::HipGrid, level::Integer = 0) = Int(ccall((:hmg_grid_cut_buffer_doubles, LIB), Int64, (Ptr{Cvoid}, Cint), g.h, level))
cut_stage_doubles(g::HipGrid) = Int(ccall((:hmg_grid_cut_stage_doubles, LIB), Int64, (Ptr{Cvoid},), g.h))
function exchange_messages(g::HipGrid, level::Integer)                   # 4 numbers per message: peer, buffer offset, count, stage offset
    n = Ref{Int64}(0)
    check(ccall((:hmg_grid_exchange_messages, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Int64, Ref{Int64}), g.h, level, C_NULL, 0, n))
    out = zeros(Int64, n[])
    check(ccall((:hmg_grid_exchange_messages, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}, Int64, Ref{Int64}), g.h, level, out, n[], n))
    reshape(out, 4, :)
end
# rehearsal of an N-rank partition on fewer GPUs (dist.PartitionedGrid(cut_owner = ...), tests and timing only)
function HipGrid(ctx::HipContext, implicit::ImplicitFineGrid{dim}, owner::Vector{Int32}, cut_owner::Vector{Int32}, rank::Integer,
                 nranks::Integer) where {dim}
    base = base_mesh(implicit)
    coords, cells = flat_nodes(base), flat_cells(base)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_grid_create_partition_rehearsal, LIB), Cint,
                (Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Cint, Cint, Ref{Ptr{Cvoid}}),
                ctx.h, dim, nlevels(implicit), nnodes(base), coords, nelements(base), cells, owner, cut_owner, rank, nranks, h))
    finalizer(g -> ccall((:hmg_grid_destroy, LIB), Cint, (Ptr{Cvoid},), g.h),
              HipGrid{dim}(h[], ctx, implicit, nlevels(implicit), nothing, NaN))
end

# threaded host setup of the checkerboard driver (driver.checkerboard_mesh / conductivity_per_element, dist.block_owner):
# the mesh of unit cubes split into simplices in the reference's order (hypercube + ∞-norm ordering,
# src/examples/homogenized_coefficients.jl:229-236), the conductivity of every element from the cube it lies in (:484-500),
# the owner rank of every cell of a brick of blocks
function checkerboard_mesh(dim::Integer, shape::Vector{Int64}, origin::Vector{Float64}; transposed_lookup::Bool = true, ordered::Bool = true)
    nn, ne = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:hmg_checkerboard_mesh_size, LIB), Cint, (Cint, Ptr{Int64}, Ref{Int64}, Ref{Int64}), dim, shape, nn, ne))
    coords, cells = zeros(Float64, dim * nn[]), zeros(Int64, (dim + 1) * ne[])
    check(ccall((:hmg_checkerboard_mesh, LIB), Cint, (Cint, Ptr{Int64}, Ptr{Float64}, Cint, Cint, Ptr{Float64}, Ptr{Int64}),
                dim, shape, origin, transposed_lookup ? 1 : 0, ordered ? 1 : 0, coords, cells))
    (reshape(coords, dim, :), reshape(cells, dim + 1, :))
end
function conductivity_per_element(dim::Integer, coords::Matrix{Float64}, cells::Matrix{Int64}, grid_shape::Vector{Int64},
                                  sigma_grid::Array{Float64}, offset::Vector{Float64})
    sigma = zeros(Float64, dim * size(cells, 2))
    check(ccall((:hmg_conductivity_per_element, LIB), Cint,
                (Cint, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                dim, size(coords, 2), coords, size(cells, 2), cells, grid_shape, sigma_grid, offset, sigma))
    reshape(sigma, dim, :)
end
function block_owner(dim::Integer, coords::Matrix{Float64}, cells::Matrix{Int64}, blocks::Vector{Int64}, width::Real, origin::Vector{Float64})
    owner = zeros(Int32, size(cells, 2))
    check(ccall((:hmg_block_owner, LIB), Cint,
                (Cint, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int64}, Float64, Ptr{Float64}, Ptr{Int32}),
                dim, size(coords, 2), coords, size(cells, 2), cells, blocks, Float64(width), origin, owner))
    owner
end

end # module
