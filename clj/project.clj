;; Adds the GPU path to raytrace-clj: the reference's own dependency vector (project.clj:6-16 of the reference)
;; plus JNA, which is the only new dependency.  Copy src/raytrace_clj/gpu.clj next to the reference's sources.
;; NOTE: written without a JVM at hand (none exists in the build container): never compiled, see INTEGRATION.md.
(defproject raytrace-clj-gpu "0.1.0-SNAPSHOT"
  :description "MI355X path for raytrace-clj's per-pixel sampling loop (librtmi.so via JNA)"
  :dependencies [[org.clojure/clojure "1.8.0"]
                 [net.mikera/imagez "0.10.0"]
                 [net.mikera/core.matrix "0.52.0"]
                 [net.mikera/vectorz-clj "0.44.0"]
                 [net.java.dev.jna/jna "5.13.0"]]
  :jvm-opts ["-Djna.library.path=../raytrace_clj_amd/lib"]
  :main ^:skip-aot raytrace-clj.gpu)
