"""MI355X (gfx950) drop-in for the reference package `rocwmma_fattn` (forward attention path)."""
