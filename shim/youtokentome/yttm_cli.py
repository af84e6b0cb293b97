"""The `yttm` entry point (reference: youtokentome/yttm_cli.py, console script `yttm=youtokentome.yttm_cli:main`)."""
from youtokentome_amd.yttm_cli import main  # noqa: F401

if __name__ == "__main__":
    main()
