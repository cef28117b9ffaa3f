"""Device-side input pipeline: Pillow-exact resize + BGR-mean conversion on the GPU, pinned double-buffered uploads."""
