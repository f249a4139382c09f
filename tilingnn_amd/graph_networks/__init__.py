"""Host-side mirror of the reference's `graph_networks` package (same module / class names and
state-dict layout), with every forward routed through libtgnn's gfx950 kernels."""
