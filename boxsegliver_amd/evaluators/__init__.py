"""Volume evaluators -- mirror of the reference's evaluators/ package."""
