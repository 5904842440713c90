"""Speech separation / enhancement models of the hot path (aps/sse): DCCRN."""
