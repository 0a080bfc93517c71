"""Drop-in entrypoint of the reference (main.py) -> lcgan_amd.main."""
from lcgan_amd.main import main, parse_args, check_args  # noqa: F401

if __name__ == "__main__":
    main()
