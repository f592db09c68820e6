"""The GUNet entry point -- drop-in for the reference's entry/main_g.py: the sub-commands liver | nf | nf2 | nf_inter with
the guided pipelines' flag groups, EvaluatorHookV2 (moving-average best checkpoint) unless --save_interval, and the plateau
hook driven by --min_delta (entry/main_g.py:55-73,145,166,171-189).  Everything else is entry/main.py."""
import sys

from .main import main as _main


def main(argv=None):
    return _main(argv, guided=True)


if __name__ == "__main__":
    sys.exit(main())
