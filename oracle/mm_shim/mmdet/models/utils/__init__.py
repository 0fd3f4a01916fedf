from .gaussian_target import gaussian_radius, gen_gaussian_target  # noqa: F401
