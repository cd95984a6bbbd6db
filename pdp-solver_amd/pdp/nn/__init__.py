"""Solver framework, plug-ins and primitives (reference: src/pdp/nn/)."""
