"""`python -m vdx.compat.run <script.py> [args...]` — run a reference strategy script with the import shims installed
(the script itself is not modified; `torchrun ... -m vdx.compat.run script.py ...` works the same way)."""
import runpy
import sys

from . import install


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m vdx.compat.run <script.py> [args...]")
    install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
