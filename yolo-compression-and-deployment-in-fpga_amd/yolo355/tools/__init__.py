"""Command-line / library tools around the engine (weight preparation)."""
