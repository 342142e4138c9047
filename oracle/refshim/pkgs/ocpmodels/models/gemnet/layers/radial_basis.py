"""RadialBasis(num_radial, cutoff, rbf={'name': 'spherical_bessel'}) -- restated in oracle/nets.py (same parameter name
rbf.frequencies, so the reference's no_weight_decay isinstance check and state_dict keys work unchanged)."""
from oracle.nets import RadialBasis, SphericalBesselBasis  # noqa: F401
