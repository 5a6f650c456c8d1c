./scripts/ubench/fetch_ubench
