"""Mirror of the reference's Metrics/ tree: device-side reductions behind the same signatures."""
