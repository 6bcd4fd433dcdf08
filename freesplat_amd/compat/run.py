"""One-command drop-in:  python -m freesplat_amd.compat.run <module> [args...]

Run from the root of an UNMODIFIED FreeSplat checkout (so that `src` is importable) with freesplat_amd on PYTHONPATH:

    python -m freesplat_amd.compat.run src.main +experiment=scannet/2views

does, in this order,
  1. compat.install()          registers `diff_gaussian_rasterization_depth` (cuda_splatting.py:5 imports it)
  2. compat.patch_reference()  imports the reference's encoder / decoder packages and rebinds the hot-path names inside
                               them (cost volume, adapter, GRU, fuse_gaussians, EncoderFreeSplat.forward, the depth tail,
                               DECODERS["splatting_cuda"]) -- before the target module builds any model
  3. runpy.run_module(<module>, run_name="__main__") with sys.argv = [<module>, args...]

`--dry-run` after the module name stops after step 2 and prints what was rebound (used by tests/test_compat_reference.py).
"""
import os
import runpy
import sys


def main(argv=None) -> dict:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        raise SystemExit(0 if argv else 2)
    module, rest = argv[0], argv[1:]
    dry = "--dry-run" in rest
    rest = [a for a in rest if a != "--dry-run"]
    if os.getcwd() not in sys.path:
        sys.path.insert(0, os.getcwd())
    from freesplat_amd import _lib, compat
    _lib.lib()                      # fail now, loudly, if libfreesplat_hip.so is missing: there is no fallback path
    compat.install()
    done = compat.patch_reference()
    for name, obj in done.items():
        print(f"[freesplat_amd] {name} -> {getattr(obj, '__module__', '?')}.{getattr(obj, '__qualname__', obj)}", file=sys.stderr)
    if dry:
        return done
    sys.argv = [module] + rest
    runpy.run_module(module, run_name="__main__", alter_sys=True)
    return done


if __name__ == "__main__":
    main()
