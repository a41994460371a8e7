# Pins the CPU oracle (oracle/gsr_oracle.c) against a LIVE run of the reference — the one thing the build
# environment cannot do (no Julia there; SURVEY.md §8c "parity unpinned": absolute image values and the order of
# equal-key instances).  A maintainer with the reference installed and any supported GPU runs
#
#     julia --project=/path/to/GaussianSplatting.jl julia/dump_reference_goldens.jl tests/golden out_dir
#
# It feeds the inputs of the three committed golden scenes (tests/golden/scene_{rgb,rgbd,rgbdn}.npz, written by the
# oracle through tests/golden/make_golden.py) to the REFERENCE rasterizer and its pullback, and writes
# `ref_scene_<mode>.npz` with image, radii, n_rendered, ranges, values_sorted (0-based), n_contrib, accum_α and the
# five gradients + ∇means_2d.  `python tools/compare_reference_dump.py tests/golden out_dir` then reports the
# differences against the committed oracle outputs with the tolerances of SURVEY.md §8(c).
#
# Array layout: numpy C-order (N,3) == Julia (3,N), so every array is permuted on the way in and out.
# The golden scenes use 64x48 / 48x40 images (multiples of 16, as rasterizer.jl:66 asserts).
using NPZ, StaticArrays, LinearAlgebra
import GaussianSplatting as GS
import GaussianSplatting: NU
import KernelAbstractions as KA

rev(x::AbstractArray) = permutedims(x, reverse(1:ndims(x)))

function main(golden_dir, out_dir)
    kab = GS.gpu_backend()
    adapt(x) = KA.allocate(kab, eltype(x), size(x)) |> y -> (copyto!(y, x); y)
    mkpath(out_dir)
    for mode in (:rgb, :rgbd, :rgbdn)
        f = npzread(joinpath(golden_dir, "scene_$(mode).npz"))
        W, H, deg = Int(f["width"]), Int(f["height"]), Int(f["sh_degree"])
        R = SMatrix{3, 3, Float32, 9}(f["R"])               # NPZ.jl already returns the logical (3,3) matrix
        t = SVector{3, Float32}(f["t"])
        intr = NU.CameraIntrinsics(nothing, SVector{2, Float32}(f["focal"]), SVector{2, Float32}(0.5f0, 0.5f0),
            SVector{2, UInt32}(W, H))
        camera = GS.Camera(R, t; intrinsics=intr, img_name="golden")
        rast = GS.GaussianRasterizer(kab; width=W, height=H, mode)
        # NPZ.jl returns arrays with the numpy (logical) shape: (N,3), (N,K,3), (N,4) -> Julia layouts (3,N), (3,K,N), (4,N)
        means = adapt(rev(f["means"])); shs = adapt(rev(f["shs"]))
        opac = adapt(reshape(f["opacities"], 1, :)); scales = adapt(rev(f["scales"])); rots = adapt(rev(f["rotations"]))
        bg = SVector{3, Float32}(f["background"])
        image = GS.rasterize(means, shs, opac, scales, rots; rast, camera, sh_degree=deg, background=bg)
        KA.synchronize(kab)
        out = Dict{String, Any}()
        out["image"] = rev(Array(image))                                   # (C,W,H) -> (H,W,C)
        out["radii"] = Array(rast.gstate.radii)[1:size(means, 2)]
        n_rendered = Int(Array(rast.gstate.points_offset)[size(means, 2)])
        out["n_rendered"] = n_rendered
        out["ranges"] = rev(Array(rast.istate.ranges))                     # (2,T) -> (T,2), 0-based [first, last+1)
        out["values_sorted"] = Array(rast.bstate.gaussian_values_sorted)[1:n_rendered] .- UInt32(1)
        out["n_contrib"] = rev(Array(rast.istate.n_contrib))
        out["accum_alpha"] = rev(Array(rast.istate.accum_α))
        vp = adapt(rev(f["vpixels"]))
        ∇ = GS.∇rasterize(vp, means, shs, scales, rots, opac, rast.gstate.radii; rast, camera, sh_degree=deg, background=bg)
        KA.synchronize(kab)
        for (name, g) in zip(("vmeans", "vshs", "vopacities", "vscales", "vrots"), ∇[1:5])
            out[name] = rev(Array(g))
        end
        out["vopacities"] = vec(out["vopacities"])
        out["vmeans2d"] = rev(reshape(reinterpret(Float32, Array(rast.gstate.∇means_2d)), 2, :))[1:size(means, 2), :]
        npzwrite(joinpath(out_dir, "ref_scene_$(mode).npz"), out)
        println("scene_$mode: n_rendered = $n_rendered (oracle: $(Int(f["n_rendered"])))")
    end
end

main(ARGS[1], ARGS[2])
