"""Run one of the reference's entry scripts UNCHANGED on this implementation:

    python -m streammos_amd.refapi /path/to/StreamMOS/val_StreamMOS.py --config config/StreamMOS.py ...

``install()`` publishes the mirror packages under the reference's import names (deep_point, point_deep,
MultiScaleDeformableAttention, deformattn, networks, models, config) before the script's first import, then the script
runs as ``__main__`` with its own argument list and its own directory at the front of ``sys.path`` (the reference's
``datasets`` / ``utils`` packages, which are outside the accelerated path, are imported from there).  The two-line edit
of INTEGRATION.md section 2 is the alternative when the script has to be started some other way (torch.distributed.run)."""
import os
import runpy
import sys


def main(argv):
    if len(argv) < 2 or argv[1] in ("-h", "--help"):
        sys.stderr.write(__doc__ + "\n")
        return 2
    script = os.path.abspath(argv[1])
    if not os.path.isfile(script):
        sys.stderr.write("streammos_amd.refapi: no such script: %s\n" % script)
        return 2
    from . import install
    install()
    sys.argv = [script] + list(argv[2:])
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
