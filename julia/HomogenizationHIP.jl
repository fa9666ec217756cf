# HomogenizationHIP.jl -- `ccall` binding of libhmg_hip.so (C ABI: include/hmg.h) for haampie/Homogenization.jl.
#
# STATUS: written against the ABI, NOT executed -- the build image has no Julia toolchain.  The same ABI is
# exercised end-to-end by the Python/ctypes mirror (homogenization.jl_amd/api.py) and its GPU parity tests.
# See INTEGRATION.md for the design; this file is the code from that document, kept loadable as a module.
module HomogenizationHIP

using Homogenization
import Homogenization: mul!, local_residual!, apply_constraint!, broadcast_interfaces!,
       zero_out_all_but_one!, restrict_to!, interpolate_and_sum_to!, smoothing_steps!, vcycle!,
       copy_to_base!, distribute!, local_rhs!, rhs_aξ∇v!, next_rhs!, LevelState, ImplicitFineGrid, L2PlusDivAGrad
import LinearAlgebra: dot, axpy!

const LIB = get(ENV, "HMG_LIB", "libhmg_hip.so")

check(rc) = rc == 0 || error(unsafe_string(ccall((:hmg_last_error, LIB), Cstring, ())))

mutable struct HipContext
    h::Ptr{Cvoid}
end
function HipContext(device::Integer = 0)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_ctx_create, LIB), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h))
    finalizer(c -> ccall((:hmg_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.h), HipContext(h[]))
end

mutable struct HipGrid
    h::Ptr{Cvoid}
    implicit::ImplicitFineGrid
end
function HipGrid(ctx::HipContext, implicit::ImplicitFineGrid{dim}) where {dim}
    base = implicit.base
    coords = collect(reinterpret(Float64, base.nodes))
    cells = collect(reinterpret(Int64, base.elements))
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_grid_create, LIB), Cint,
          (Ptr{Cvoid}, Cint, Cint, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ref{Ptr{Cvoid}}),
          ctx.h, dim, implicit.levels, length(base.nodes), coords, length(base.elements), cells, h))
    finalizer(g -> ccall((:hmg_grid_destroy, LIB), Cint, (Ptr{Cvoid},), g.h), HipGrid(h[], implicit))
end
set_operator!(g::HipGrid, A::L2PlusDivAGrad) =
    check(ccall((:hmg_grid_set_operator, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64),
          g.h, collect(reinterpret(Float64, A.σs)), A.λ))
shrink!(g::HipGrid, ncells::Integer, nnodes::Integer) =
    check(ccall((:hmg_grid_shrink, LIB), Cint, (Ptr{Cvoid}, Int64, Int64), g.h, ncells, nnodes))

mutable struct HipMatrix <: AbstractMatrix{Float64}
    h::Ptr{Cvoid}
    grid::HipGrid
    level::Int
end
function HipMatrix(g::HipGrid, level::Integer)
    h = Ref{Ptr{Cvoid}}()
    check(ccall((:hmg_vec_create, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), g.h, level, h))
    finalizer(v -> ccall((:hmg_vec_destroy, LIB), Cint, (Ptr{Cvoid},), v.h), HipMatrix(h[], g, level))
end
Base.size(v::HipMatrix) = (Int(ccall((:hmg_grid_nf, LIB), Int64, (Ptr{Cvoid}, Cint), v.grid.h, v.level)),
                           Int(ccall((:hmg_grid_ncells, LIB), Int64, (Ptr{Cvoid},), v.grid.h)))
Base.getindex(v::HipMatrix, i...) = error("HipMatrix lives in HBM: use Array(v) to download")
Base.copyto!(v::HipMatrix, a::Matrix{Float64}) =
    (check(ccall((:hmg_vec_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), v.h, a)); v)
function Base.Array(v::HipMatrix)
    a = Matrix{Float64}(undef, size(v)...)
    check(ccall((:hmg_vec_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), v.h, a))
    a
end
Base.fill!(v::HipMatrix, x) = (check(ccall((:hmg_vec_fill, LIB), Cint, (Ptr{Cvoid}, Float64), v.h, x)); v)
Base.copyto!(d::HipMatrix, s::HipMatrix) =
    (check(ccall((:hmg_vec_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), d.h, s.h)); d)
function dot(x::HipMatrix, y::HipMatrix)
    o = Ref(0.0)
    check(ccall((:hmg_vec_dot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}), x.h, y.h, o))
    o[]
end
axpy!(a, x::HipMatrix, y::HipMatrix) =
    (check(ccall((:hmg_vec_axpy, LIB), Cint, (Float64, Ptr{Cvoid}, Ptr{Cvoid}), a, x.h, y.h)); y)
function norm_unique(r::HipMatrix)      # norm(r) after zero_out_all_but_one!(r), r untouched
    o = Ref(0.0)
    check(ccall((:hmg_vec_norm_unique, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), r.h, o))
    o[]
end

LevelState(g::HipGrid, level::Integer) = LevelState{Float64,HipMatrix}((HipMatrix(g, level) for _ in 1:5)...)

mul!(α::Float64, base, A::L2PlusDivAGrad, x::HipMatrix, y::HipMatrix) =
    check(ccall((:hmg_apply, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Ptr{Cvoid}), x.grid.h, x.level, α, x.h, y.h))
local_residual!(implicit, A::L2PlusDivAGrad, c::LevelState{Float64,HipMatrix}, k::Int) =
    check(ccall((:hmg_residual, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), c.x.grid.h, k, c.x.h, c.b.h, c.r.h))
apply_constraint!(x::HipMatrix, level::Int, z, implicit) =
    check(ccall((:hmg_constraint, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), x.grid.h, level, x.h))
broadcast_interfaces!(x::HipMatrix, implicit, level::Int) =
    check(ccall((:hmg_interface_sum, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), x.grid.h, level, x.h))
zero_out_all_but_one!(x::HipMatrix, implicit, level::Int) =
    check(ccall((:hmg_zero_duplicates, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), x.grid.h, level, x.h))
restrict_to!(y::HipMatrix, P, x::HipMatrix) =
    check(ccall((:hmg_restrict, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}), x.grid.h, x.level, x.h, y.h))
interpolate_and_sum_to!(y::HipMatrix, P, x::HipMatrix) =
    check(ccall((:hmg_prolong_add, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}), y.grid.h, y.level, x.h, y.h))
copy_to_base!(u::Vector{Float64}, v::HipMatrix, implicit) =
    check(ccall((:hmg_gather_base, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), v.grid.h, v.h, u))
distribute!(v::HipMatrix, u::Vector{Float64}, implicit) =
    check(ccall((:hmg_scatter_base, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), v.grid.h, u, v.h))
smoothing_steps!(steps::Integer, implicit, ops, c::LevelState{Float64,HipMatrix}, k::Int) =
    check(ccall((:hmg_smooth, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
          c.x.grid.h, k, steps, c.x.h, c.b.h, c.r.h, c.p.h, c.Ap.h))
function vcycle!(implicit, base, ops, levels::Vector{LevelState{Float64,HipMatrix}}, k::Int, steps = 2)
    hs = Ptr{Cvoid}[getfield(l, f).h for l in levels for f in (:x, :b, :r, :p, :Ap)]
    check(ccall((:hmg_vcycle, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Ptr{Cvoid}}), levels[1].x.grid.h, k, steps, 2, hs))
end

# driver right-hand sides (src/implicit_fine_grid.jl:391, src/examples/homogenized_coefficients.jl:449,695)
local_rhs!(b::HipMatrix, implicit) = check(ccall((:hmg_local_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), b.grid.h, b.h))
rhs_aξ∇v!(b::HipMatrix, ∂ϕs, implicit, σs, ξ) =
    check(ccall((:hmg_rhs_axi_grad, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Cvoid}), b.grid.h, collect(Float64, ξ), b.h))
next_rhs!(b::HipMatrix, x::HipMatrix, implicit, ops) =
    check(ccall((:hmg_next_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), b.grid.h, x.h, b.h))

end # module
