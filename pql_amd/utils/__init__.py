"""Host-side helpers: config loader, plugin registry, noise, running statistics, logging."""
