"""Module-name parity with team_code_transfuser/latentTF.py."""
from .transfuser import latentTFBackbone  # noqa: F401
