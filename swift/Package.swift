// swift-tools-version: 5.7
// The reference's manifest (Package.swift:1-84 of lukevanin/SIFTMetal) with its Metal seam replaced by the MI355X one:
// the `MetalShaders` Clang-module target (Sources/MetalShaders/module.modulemap:1-4, the C structs shared with the .metal
// files) gives way to the system-library target `CSiftmi` = include/siftmi.h + libsiftmi.so, and `SIFT` is implemented by
// Sources/SIFTMetal/SIFT/SIFT+MI355X.swift instead of SIFT.swift / DifferenceOfGaussians.swift / SIFTOctave.swift and the
// `Metal Compute` wrappers.  The value types (SIFTKeypoint, SIFTDescriptor, SIFTCorrespondence, IntegralSize, IntVector,
// FloatVector) stay the reference's own files.
//
// Not compiled in the build image (no Swift toolchain there); the same boundary is exercised by the ctypes binding in
// siftmetal_amd/ which mirrors it call for call.  Build on a host with Swift and ROCm:
//   swift build -Xcc -I<repo>/include -Xlinker -L<repo>/siftmetal_amd -Xlinker -rpath -Xlinker <repo>/siftmetal_amd

import PackageDescription

let package = Package(
    name: "SIFTMetal",
    products: [
        .library(name: "SIFTMetal", targets: ["SIFTMetal"]),
    ],
    targets: [
        .systemLibrary(name: "CSiftmi", path: "Sources/CSiftmi"),
        .target(
            name: "SIFTMetal",
            dependencies: ["CSiftmi"],
            path: "Sources/SIFTMetal",
            swiftSettings: [.unsafeFlags(["-O"])],
            linkerSettings: [.linkedLibrary("siftmi")]
        ),
    ]
)
