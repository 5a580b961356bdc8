# SSQPHip.jl -- Julia side of the drop-in boundary (UNVERIFIED: no Julia in the build image or on the GPU box).
#
# Adds GPU methods for the two reference entry points of the hot path and leaves everything else
# (QP/LP types, Settings, MOI wrapper, LP solvers) to StatusSwitchingQP.jl itself:
#
#   solveQP(Q::QP{Float64}, S, x0; settings)      replaces src/SSQP.jl:237-377
#   solveQP(Q::QP{Float64}; settings, settingsLP) replaces src/SSQP.jl:224-234
#
# Usage:  ENV["SSQP_HIP_LIB"] = "/path/to/libssqp_hip.so"; include("SSQPHip.jl"); using .SSQPHip
#         z, S, status = SSQPHip.solveQP(Q)          # same return triple as the reference
module SSQPHip

using StatusSwitchingQP
using StatusSwitchingQP: QP, Settings, Status, DN

export solveQP

const libssqp = get(ENV, "SSQP_HIP_LIB", "libssqp_hip.so")

# struct ssqp_settings { int32 maxIter; int32 rule; double tol; double tolG; }   (include/ssqp_hip.h)
struct CSettings
    maxIter::Int32
    rule::Int32
    tol::Float64
    tolG::Float64
end
function CSettings(s::Settings{Float64})
    s.rule == :Dantzig || error("SSQPHip: only rule=:Dantzig is implemented for Phase-1")
    CSettings(Int32(s.maxIter), Int32(0), s.tol, s.tolG)
end

const _ctx = Ref{Ptr{Cvoid}}(C_NULL)
function ctx()
    if _ctx[] == C_NULL
        rc = ccall((:ssqp_ctx_create, libssqp), Cint, (Cint, Ref{Ptr{Cvoid}}), 0, _ctx)
        rc == 0 || error("ssqp_ctx_create failed with code $rc (2 = no HIP device; there is no CPU fallback)")
    end
    _ctx[]
end

check(rc) = rc == 0 || error("libssqp_hip: code $rc: " *
    unsafe_string(ccall((:ssqp_last_error, libssqp), Cstring, (Ptr{Cvoid},), _ctx[])))

# detail codes: 1/2 = a cholesky in the loop would have thrown, 3 = lu in Phase-1 would have thrown
rethrow_detail(detail) = detail in (1, 2) ? throw(LinearAlgebra.PosDefException(detail)) :
                         detail == 3 ? throw(LinearAlgebra.SingularException(0)) : nothing

"""
    solveQP(Q::QP{Float64}, S::Vector{Status}, x0; settings=Settings{Float64}())

The hot path on the GPU.  `S` is mutated in place and returned, `x0` is not modified (SSQP.jl:264-266).
"""
function solveQP(Q::QP{Float64}, S::Vector{Status}, x0::Vector{Float64}; settings=Settings{Float64}(),
                 throw_like_reference::Bool=false)
    length(S) == Q.N + Q.J || throw(DimensionMismatch("S must have length N+J"))
    z = Vector{Float64}(undef, Q.N)
    status = Ref{Int64}(0)
    detail = Ref{Int32}(0)
    cs = Ref(CSettings(settings))
    Si = reinterpret(Int32, S)                       # @enum Status is Int32: IN=0 DN=1 UP=2 OE=3 EO=4
    GC.@preserve Q S x0 z begin
        check(ccall((:ssqp_solve_f64, libssqp), Cint,
            (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ref{CSettings},
             Ref{Int64}, Ref{Int32}),
            ctx(), Q.N, Q.M, Q.J, Q.V, Q.A, Q.G, Q.q, Q.b, Q.g, Q.d, Q.u, Si, x0, z, cs, status, detail))
    end
    throw_like_reference && rethrow_detail(detail[])
    return z, S, Int(status[])
end

"""
    solveQP(Q::QP{Float64}; settings=Settings{Float64}(), settingsLP=settings)

Phase-1 (initQP) on the host inside the library, then the GPU loop.
"""
function solveQP(Q::QP{Float64}; settings=Settings{Float64}(), settingsLP=settings, throw_like_reference::Bool=false)
    if Q.mc <= 0
        return zeros(Float64, Q.N), fill(DN, Q.N), -1          # SSQP.jl:226-228
    end
    S = Vector{Status}(undef, Q.N + Q.J)
    z = Vector{Float64}(undef, Q.N)
    status = Ref{Int64}(0)
    detail = Ref{Int32}(0)
    cs = Ref(CSettings(settings))
    csl = Ref(CSettings(settingsLP))
    Si = reinterpret(Int32, S)
    GC.@preserve Q S z begin
        check(ccall((:ssqp_solve_full_f64, libssqp), Cint,
            (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Int32}, Ptr{Float64}, Ref{CSettings},
             Ref{CSettings}, Ref{Int64}, Ref{Int32}),
            ctx(), Q.N, Q.M, Q.J, Q.V, Q.A, Q.G, Q.q, Q.b, Q.g, Q.d, Q.u, Q.mc, Si, z, cs, csl, status, detail))
    end
    throw_like_reference && rethrow_detail(detail[])
    return z, S, Int(status[])
end

# BigFloat (and any T != Float64) stays on the reference's CPU path
solveQP(Q::QP, args...; kwargs...) = StatusSwitchingQP.solveQP(Q, args...; kwargs...)

end # module
