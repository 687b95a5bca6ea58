"""CPU oracle for the ray-tracing hot path — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(); the product
package (vulkan_raytracing_amd) never imports it.
"""
