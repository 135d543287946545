"""Host-side utilities mirroring pgl/utils (only what the message-passing path needs)."""
