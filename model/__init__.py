"""Drop-in module path of the reference (model.*): every module re-exports its stove_amd counterpart."""
