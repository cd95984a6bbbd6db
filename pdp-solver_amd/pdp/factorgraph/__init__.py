"""Input pipeline and predict driver (reference: src/pdp/factorgraph/)."""
from pdp.factorgraph.dataset import FactorGraphDataset  # noqa: F401
from pdp.factorgraph.base import FactorGraphTrainerBase  # noqa: F401
