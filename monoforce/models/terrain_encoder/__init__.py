from .lss import LiftSplatShoot  # noqa: F401
