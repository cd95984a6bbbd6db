"""Input pipeline and predict driver (reference: src/pdp/factorgraph/)."""
from pdp.factorgraph.dataset import FactorGraphDataset  # noqa: F401
